// gru_train_kernels.hip -- training step of the GRU head (reference gru/gru_train.py:232-249) on gfx950.
//
//   forward  : the inference layer kernels with the per-step activations (r, z, n, gh_n, h_t) saved row-major
//   loss     : target = [y | |out[:, :C/2].detach() - y|], MSE over all B*C entries (gru_train.py:237-245), d(loss)/d(out)
//   head bwd : d(pre) = dout * out(1-out); d(fc.weight), d(fc.bias), d(h_T)
//   sweep    : per layer, reverse in time: gate derivatives on the VALU, then
//                dx_t      = [da_r | da_z | da_n    ] . W_ih      (fp32 MFMA, reduction over the 3H gate units)
//                dh_{t-1} += [da_r | da_z | da_n * r] . W_hh
//              with dh carried in LDS, gate derivatives staged in LDS as MFMA A-fragments and written out once
//              ([T][B][3H]) for the weight gradients
//   weights  : dW_ih = sum_t dGI_t^T X_t, dW_hh = sum_t dGH_t^T H_{t-1} (T*B-long reductions): split-K fp32 MFMA kernel with
//              both operands coalesced straight from memory and fp32 atomics for the partial tiles (dw_kernel); the bias
//              gradients ride along / a column-sum kernel.
// All gradients land in one flat fp32 vector in the flat parameter layout (include/optistate_hip.h), which is the
// single bucket the data-parallel step all-reduces over RCCL.
#include "launch.hpp"

#include <stdlib.h>

#include <type_traits>

#include "gru_common.hpp"
#include "kf_device.hpp"   // buffer addressing helpers
#include "gru_device.hpp"  // OSL_TS (development timestamps)

namespace ost {

using osg::sigmoidf_;

// ---- loss + d(out) -------------------------------------------------------------------------------------------
// out [B][C], y [B][C/2]; target [B][C] (optional), dout [B][C]; loss: one float, written (not accumulated) by the LAST
// workgroup to finish, which adds the per-workgroup partial sums in index order: no memset in front of the launch (4.5 us
// of a 2.2 ms training step) and a loss that does not depend on the order the workgroups retire in.
// scratch: [LOSS_MAXBLK] partial sums + one ticket counter (zero before the first launch; atomicInc wraps it back to zero).
constexpr int LOSS_MAXBLK = 128;
__global__ __launch_bounds__(256) void loss_kernel(int B, int C, const float *out, const float *y, float *target, float *dout, float *loss,
                                                   float *scratch)
{
    __shared__ float part[4];
    __shared__ bool last;
    const int half = C / 2, n = B * C;
    const float inv = 1.0f / (float)n;
    float sq = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = i / C, c = i % C;
        const float o = out[i];
        const float tgt = c < half ? y[b * half + c] : fabsf(out[b * C + (c - half)] - y[b * half + (c - half)]);
        const float d = o - tgt;
        sq += d * d;
        if (target) target[i] = tgt;
        dout[i] = 2.0f * d * inv;                  // the target is detached (gru_train.py:239): no gradient through |.|
    }
    // wave reduction, workgroup reduction, ONE atomic per workgroup (an atomic per wave put 3,072 of them on one address at
    // the training batch: 41 us for a 197 k-element pass)
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&scratch[blockIdx.x], (part[0] + part[1] + part[2] + part[3]) * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned ticket = __hip_atomic_fetch_add((unsigned *)(scratch + LOSS_MAXBLK), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    float v = 0.f;
    if (threadIdx.x < 64) {                  // gridDim.x <= LOSS_MAXBLK = 2 x 64: fixed order, fixed tree
        const unsigned i0 = threadIdx.x, i1 = threadIdx.x + 64;
        if (i0 < gridDim.x) v = __hip_atomic_load(&scratch[i0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i1 < gridDim.x) v += __hip_atomic_load(&scratch[i1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) {
            *loss = v;
            __hip_atomic_store((unsigned *)(scratch + LOSS_MAXBLK), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// hardware float add at the L2 (atomicAdd(float *) without -munsafe-fp-atomics is a compare-and-swap loop, which under the
// contention of B/64 workgroups on 3,096 addresses took most of head_backward_kernel's 45 us)
__device__ __forceinline__ void global_fadd(float *base, uint32_t bytes, uint32_t index, float v)
{
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, osk::make_rsrc(base, bytes), index * 4u, 0u, 0);
}

// ---- head backward -------------------------------------------------------------------------------------------
// dpre = dout * out * (1 - out) (sigmoid) or dout; dh_T [B][H] = dpre . fcw; dfcw [C][H] += dpre^T h_T; dfcb += sum dpre.
// One workgroup per 64 rows, 512 threads = (hidden unit k: 128 at a time) x (4 quarters).  Register blocking: for dh_T a
// thread holds its column of fcw (C values) and walks its 16 rows with the row's dpre broadcast from LDS; for dfcw it holds
// C/4 accumulators and walks the 64 rows.  (Two halves on 256 threads: 21.7 us at the training batch, one wave per SIMD
// with nothing to cover its LDS reads.)  (The first version did one LDS/global read per multiply-add and took 69 us at
// the training batch.)  Partial dfcw / dfcb are added with float atomics: C*H + C addresses, B/64 contributions each.
constexpr int HB_CMAX = 32;            // classes per pass, held in registers (the LDS tile is zero-padded to a multiple of it:
                                       // straight-line inner loops -- with run-time class bounds hipcc put an
                                       // s_waitcnt vmcnt(0) behind every LDS read and each row waited for the previous row's store)
__global__ __launch_bounds__(512) void head_backward_kernel(int B, int H, int C, int use_sigmoid, const float *out, const float *dout,
                                     const float *hT /*[B][H]*/, const float *fcw, float *dhT /*[B][H]*/, float *dfcw,
                                     float *dfcb)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];          // dpre [64][Cp] , h [64][H+1]
    const int Cp = (C + HB_CMAX - 1) / HB_CMAX * HB_CMAX;
    float *dp = sm, *hs = sm + 64 * Cp;
    const int row0 = blockIdx.x * 64;
    // staging loops unrolled eight deep with range-checked buffer loads: rolled, every iteration was one HBM round trip
    // (load - wait - LDS store), 32 of them in a row for the h tile
    const osk::rsrc_t ro = osk::make_rsrc(out, (uint32_t)((size_t)B * C * 4)), rd = osk::make_rsrc(dout, (uint32_t)((size_t)B * C * 4)),
                      rh = osk::make_rsrc(hT, (uint32_t)((size_t)B * H * 4));
#pragma unroll 8
    for (int i = threadIdx.x; i < 64 * Cp; i += 512) {
        const int r = i / Cp, c = i % Cp;
        const uint32_t off = c < C ? (uint32_t)(((size_t)(row0 + r) * C + c) * 4) : 0xffffffffu;      // pad columns: out of range -> 0
        const float o = osk::buf_load(ro, off, 0u), dv = osk::buf_load(rd, off, 0u);
        dp[i] = dv * (use_sigmoid ? o * (1.0f - o) : 1.0f);
    }
#pragma unroll 8
    for (int i = threadIdx.x; i < 64 * H; i += 512) {
        const int r = i / H, k = i % H;
        hs[r * (H + 1) + k] = osk::buf_load(rh, (uint32_t)(((size_t)(row0 + r) * H + k) * 4), 0u);   // rows past B: 0
    }
    __syncthreads();
    const int qtr = threadIdx.x >> 7, kk = threadIdx.x & 127;
    for (int k = kk; k < H; k += 128) {
        for (int c0 = 0; c0 < Cp; c0 += HB_CMAX) {
            // dh_T: rows qtr*16 .. +16
            float w[HB_CMAX];
#pragma unroll
            for (int c = 0; c < HB_CMAX; c++) w[c] = c0 + c < C ? fcw[(size_t)(c0 + c) * H + k] : 0.f;
            const osk::rsrc_t rdh = osk::make_rsrc(dhT, (uint32_t)((size_t)B * H * 4));      // rows past B: dropped by the range check
#pragma unroll 4
            for (int r = qtr * 16; r < qtr * 16 + 16; r++) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < HB_CMAX; c += 4) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(&dp[r * Cp + c0 + c]);     // wave-uniform: broadcast
                    s = fmaf(d4.x, w[c], s); s = fmaf(d4.y, w[c + 1], s); s = fmaf(d4.z, w[c + 2], s); s = fmaf(d4.w, w[c + 3], s);
                }
                const uint32_t off = (uint32_t)(((size_t)(row0 + r) * H + k) * 4);
                if (c0 == 0) osk::buf_store(rdh, off, 0u, s);
                else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(s, rdh, off, 0u, 0);
            }
            // dfcw: classes c0 + qtr*8 .. +8 over all 64 rows
            float acc[HB_CMAX / 4];
#pragma unroll
            for (int c = 0; c < HB_CMAX / 4; c++) acc[c] = 0.f;
            const int cb = c0 + qtr * (HB_CMAX / 4);
#pragma unroll 4
            for (int r = 0; r < 64; r++) {
                const float hv = hs[r * (H + 1) + k];
#pragma unroll
                for (int c = 0; c < HB_CMAX / 4; c += 4) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(&dp[r * Cp + cb + c]);
                    acc[c] = fmaf(d4.x, hv, acc[c]); acc[c + 1] = fmaf(d4.y, hv, acc[c + 1]);
                    acc[c + 2] = fmaf(d4.z, hv, acc[c + 2]); acc[c + 3] = fmaf(d4.w, hv, acc[c + 3]);
                }
            }
#pragma unroll
            for (int c = 0; c < HB_CMAX / 4; c++)
                if (cb + c < C) global_fadd(dfcw, (uint32_t)(C * H) * 4u, (uint32_t)((cb + c) * H + k), acc[c]);
        }
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < 64; r++) s += dp[r * Cp + c];
        global_fadd(dfcb, (uint32_t)C * 4u, (uint32_t)c, s);
    }
}

// ---- transposed weight packing for the backward MFMAs --------------------------------------------------------
// W [3H][K] row-major -> for each 32-wide output chunk c (columns of W) and each gate-unit pair q: 64 floats,
// lane l -> W[2q + (l>>5)][c*32 + (l&31)] (0 beyond K).
// every layer's W_ih and W_hh in ONE launch (blockIdx.z = matrix): eight ~20 us launches per backward otherwise
struct PackTAll {
    int n;
    int K[32], chunks[32];
    const float *W[32];
    float *dst[32];
};
// The launch is the first of a backward pass, so it also clears the flat gradient vector the later kernels add into (a
// separate hipMemsetAsync cost a 4.5 us launch of its own).
__global__ void pack_T_all_kernel(const PackTAll a, int H3, float *zero, size_t nzero)
{
    {
        const size_t nth = (size_t)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
        const size_t tid = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        for (size_t i = tid; i < nzero; i += nth) zero[i] = 0.f;
    }
    const int m = blockIdx.z;
    if (m >= a.n || (int)blockIdx.x >= a.chunks[m]) return;
    const int K = a.K[m], chunk = blockIdx.x, n = (H3 / 2) * 64;
    const float *W = a.W[m];
    float *d = a.dst[m] + (size_t)chunk * n;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        const int lane = i & 63, q = i >> 6;
        const int j = 2 * q + (lane >> 5), col = chunk * 32 + (lane & 31);
        d[i] = col < K ? W[(size_t)j * K + col] : 0.f;
    }
}

struct SweepArgs {
    int B, T, K, H;
    int need_dx;                 // write dx (layer > 0 or caller wants d(input))
    const float *sv_r, *sv_z, *sv_n, *sv_g, *sv_h;   // [T][B][H]
    const float *dy;             // [T][B][H] gradient w.r.t. this layer's outputs, or null
    const float *dy_last;        // [B][H] gradient w.r.t. h_T only (top layer), or null
    const float *wihT, *whhT;    // transposed-packed weights
    float *dg4;                  // [T][B][4H]: da_r | da_z | da_n | da_n * r  (W_ih's products read sections 0-2, W_hh's 0, 1, 3)
    float *dx;                   // [T][B][K]
    // bwd_sweep_stack_kernel only (null otherwise): progress counters of the layer above (its dx is this layer's dy) and of this
    // workgroup; a counter holds the number of steps published, T - t after step t
    const uint32_t *flag_prev;
    uint32_t *flag_mine;
    int32_t *err = nullptr;      // the context's error word (launch.hpp): bit 0 = a bounded wait expired
    int32_t *err_local = nullptr;    // its twin in device memory
    uint32_t max_polls = 1u << 22;
    int drop_from = -1;          // tests (OS_STACK_DBG_DROP): stop publishing once T - 1 - t reaches this count of steps done
};

// RB = 32-row blocks per workgroup (BM = 32*RB).  LDS: dh [BM][H+1] | dG [BM][4H+1] (sections da_r, da_z, da_n, da_n*r).
// NW = wavefronts per workgroup.  The sweep is serial in T and a workgroup owns its rows, so at the training batch (8192
// rows = one 32-row workgroup per CU) nothing else can hide a wave's stalls: eight waves (two per SIMD) split the gate
// derivatives and the output chunks finer and cover each other's load / LDS latency.
// WR = leading k-pairs of a wave's weight chunk held in REGISTERS for the whole launch (needs one work item per wave).
// Vector-memory loads return in order per wave, so the first weight fragments of a step's MFMA phase used to queue behind
// the 48 activation prefetches issued just before them and the phase started one HBM latency late (1.75 us of every step);
// with 32 resident k-pairs the first 32 MFMAs need nothing from memory and that latency is covered.
// STACK (bwd_sweep_stack_kernel below): all layers of a small batch in ONE launch, blockIdx.y = 0 for the top layer; layer l takes
// dy[t] = dx_{l+1}[t] as soon as the layer above has published step t (both run t = T-1 .. 0).  Same protocol as gru_stack_kernel
// (gru_kernels.hip): write-through stores, every wave drains its stores, workgroup barrier, agent-scope release of the counter;
// the consumer polls with agent-scope acquires (bounded: a lost producer poisons dy with NaN) and reads with sc0 sc1 loads.
template <int RB, int NW, int WR, bool STACK>
__device__ __forceinline__ void sweep_body(const SweepArgs &a)
{
    static_assert(WR == 0 || (RB == 1 && WR % 16 == 0), "resident weights: one row block, whole 16-k-pair blocks");
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, K = a.K, HS = H + 1, GS = 4 * H + 1, BM = 32 * RB;
    float *dh = sm, *dG = sm + BM * HS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * BM;
    const size_t B = (size_t)a.B;
    for (int i = threadIdx.x; i < BM * HS; i += NT) dh[i] = 0.f;
    __syncthreads();
    const int nchx = a.need_dx ? (K + 31) / 32 : 0, nchh = H / 32;
    const int Q = 3 * H / 2;                   // gate-unit pairs in the reduction

    // Saved activations of a step, prefetched into registers: the loads of step t-1 are issued right after the barrier that
    // ends step t's gate-derivative phase and fly underneath its MFMA phase (PMC on the 8192-window step: the waves of this
    // kernel spent 44 % of their cycles waiting, most of it at the top of every step for these six streams).  Read-once /
    // write-once streams are non-temporal so that they do not evict the L2-resident weights.
    constexpr int EL = 32 * RB * 128 / NT;             // elements per thread at H = 128 (fewer iterations for smaller H)
#ifndef OS_SWEEP_PF
#define OS_SWEEP_PF 1
#endif
    constexpr bool PF = OS_SWEEP_PF && EL <= 8;        // prefetch across the MFMA phase where it fits the register budget (eight waves)
    const int lgH = 31 - __builtin_clz(H);             // H is 32, 64 or 128: shifts instead of integer division
    const int nel = BM * H;
    constexpr int ELP = PF ? EL : 1;
    float pf_r[ELP], pf_z[ELP], pf_n[ELP], pf_g[ELP], pf_h[ELP], pf_d[ELP];
    // buffer addressing: one wave-uniform descriptor per stream, ONE 32-bit offset register per element shared by the six
    // streams (flat 64-bit addresses cost two registers per load in flight: 96 here, which spilled)
    uint32_t pf_off[ELP];
#pragma unroll
    for (int e = 0; e < ELP; e++) {
        const int i = threadIdx.x + e * NT, ic = i < nel ? i : 0;
        const int r = ic >> lgH, c = ic & (H - 1), g = row0 + r;
        pf_off[e] = (uint32_t)(((size_t)(g < a.B ? g : a.B - 1) * H + c) * 4);
    }
    bool lost = STACK && osg::stack_lost_already(a.err_local);      // STACK: latched after one expired wait anywhere in the launch (no further waits)
    auto prefetch = [&](int t) {
        const uint32_t step = (uint32_t)((size_t)t * B * H * 4), bytes = (uint32_t)((size_t)a.T * B * H * 4);
        const osk::rsrc_t rr_ = osk::make_rsrc(a.sv_r, bytes), rz_ = osk::make_rsrc(a.sv_z, bytes), rn_ = osk::make_rsrc(a.sv_n, bytes),
                          rg_ = osk::make_rsrc(a.sv_g, bytes), rh_ = osk::make_rsrc(a.sv_h, bytes),
                          rd_ = osk::make_rsrc(a.dy ? a.dy : a.sv_r, bytes);
        const uint32_t hstep = t > 0 ? step - (uint32_t)(B * H * 4) : 0u;
#pragma unroll
        for (int e = 0; e < ELP; e++) {
            pf_r[e] = osk::buf_load_nt(rr_, pf_off[e], step); pf_z[e] = osk::buf_load_nt(rz_, pf_off[e], step);
            pf_n[e] = osk::buf_load_nt(rn_, pf_off[e], step); pf_g[e] = osk::buf_load_nt(rg_, pf_off[e], step);
            pf_h[e] = osk::buf_load_nt(rh_, pf_off[e], hstep);
            if (!(STACK && a.flag_prev)) pf_d[e] = osk::buf_load_nt(rd_, pf_off[e], step);
        }
        if constexpr (STACK) {
            if (a.flag_prev) {                                 // dy[t] comes from the layer above inside this launch
                if (!lost) {
                    lost = true;
                    for (uint32_t spin = 0; spin < a.max_polls; spin++) {
                        if (__hip_atomic_load(a.flag_prev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)(a.T - t)) { lost = false; break; }
                        if ((spin & 1023u) == 1023u && osg::stack_lost_already(a.err_local)) break;      // somebody else gave up: so do we
                        __builtin_amdgcn_s_sleep(4);
                    }
                    // reported in the context's error word: the call (or the next one) fails with -20, adam_kernel skips its update
                    if (lost && lane == 0) {
                        if (a.err_local) __hip_atomic_fetch_or(a.err_local, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.err) __hip_atomic_fetch_or(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
#pragma unroll
                for (int e = 0; e < ELP; e++) {
                    pf_d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd_, pf_off[e], step, 17));
                    if (lost) pf_d[e] = __builtin_nanf("");
                }
            }
        }
    };
    if (PF) prefetch(a.T - 1);

    // work items: output chunks; with need_dx == 0 (layer 0: only the four recurrent chunks) every chunk's reduction is
    // split in two halves so that all eight waves have work, and the halves meet in dh through LDS atomics
    const int qsplit = (!a.need_dx && NW > nchh) ? 2 : 1;
    float wreg[WR > 0 ? WR : 1];
    if constexpr (WR > 0) {
        const int item = wave < (nchx + nchh) * qsplit ? wave : 0;
        const int ch = item / qsplit, qh = item % qsplit;
        const bool is_h = ch >= nchx;
        const float *wp = (is_h ? a.whhT : a.wihT) + (size_t)(is_h ? ch - nchx : ch) * Q * 64 + (size_t)qh * (Q / qsplit) * 64;
#pragma unroll
        for (int d = 0; d < WR; d++) wreg[d] = wp[d * 64 + lane];
    }

    for (int t = a.T - 1; t >= 0; t--) {
        if constexpr (PF) {
        // ---- gate derivatives (VALU), coalesced over the hidden index ----
        {
            const float *__restrict__ pdl = a.dy_last;
            const bool last = (t == a.T - 1);
            const uint32_t gbytes = (uint32_t)((size_t)a.T * B * 4 * H * 4), gstep = (uint32_t)((size_t)t * B * 4 * H * 4), gH = (uint32_t)H * 4u;
            const osk::rsrc_t rg4_ = osk::make_rsrc(a.dg4, gbytes);
#pragma unroll
            for (int e = 0; e < ELP; e++) {
                const int i = threadIdx.x + e * NT;
                if (i >= nel) continue;
                const int r = i >> lgH, c = i & (H - 1), g = row0 + r;
                const bool ok = g < a.B;
                const int gc = ok ? g : a.B - 1;
                const float rr = pf_r[e], zz = pf_z[e], nn = pf_n[e], gg = pf_g[e], hp = t > 0 ? pf_h[e] : 0.f;
                float dht = dh[r * HS + c] + (a.dy ? pf_d[e] : 0.f);
                if (pdl && last) dht += pdl[(size_t)gc * H + c];
                if (!ok) dht = 0.f;
                const float dn = dht * (1.0f - zz);
                const float dz = dht * (hp - nn);
                const float dhc = dht * zz;
                const float dan = dn * (1.0f - nn * nn);
                const float danr = dan * rr;
                const float dar = dan * gg * rr * (1.0f - rr);
                const float daz = dz * zz * (1.0f - zz);
                if (ok) {
                    // buffer stores: the element's offset register is shared by the six streams (gate thirds through the SGPR
                    // offset); flat 64-bit addresses for 48 stores in flight cost ~100 registers and spilled the prefetch
                    const uint32_t og = (uint32_t)(((size_t)g * 4 * H + c) * 4);
                    osk::buf_store_nt(rg4_, og, gstep, dar); osk::buf_store_nt(rg4_, og, gstep + gH, daz);
                    osk::buf_store_nt(rg4_, og, gstep + 2 * gH, dan); osk::buf_store_nt(rg4_, og, gstep + 3 * gH, danr);
                }
                dG[r * GS + c] = dar; dG[r * GS + H + c] = daz; dG[r * GS + 2 * H + c] = dan; dG[r * GS + 3 * H + c] = danr;
                dh[r * HS + c] = dhc;            // the z * dh part of dh_{t-1}; the matrix part is added below
            }
        }
        } else {
        // ---- gate derivatives (VALU), coalesced over the hidden index.  Unrolled 8x with restrict-qualified streams so
        // that the loads of a batch are in flight together; read-once / write-once streams are non-temporal so they do
        // not evict the L2-resident weights ----
        {
            const float *__restrict__ pr = a.sv_r, *__restrict__ pz = a.sv_z, *__restrict__ pn = a.sv_n,
                        *__restrict__ pg = a.sv_g, *__restrict__ ph = a.sv_h, *__restrict__ pdy = a.dy,
                        *__restrict__ pdl = a.dy_last;
            float *__restrict__ og4 = a.dg4;
            const bool last = (t == a.T - 1);
            #pragma unroll 8
            for (int i = threadIdx.x; i < BM * H; i += NT) {
                const int r = i >> lgH, c = i & (H - 1), g = row0 + r;
                const bool ok = g < a.B;
                const int gc = ok ? g : a.B - 1;
                const size_t o = ((size_t)t * B + gc) * H + c;
                const float rr = __builtin_nontemporal_load(pr + o), zz = __builtin_nontemporal_load(pz + o),
                            nn = __builtin_nontemporal_load(pn + o), gg = __builtin_nontemporal_load(pg + o);
                const float hp = t > 0 ? __builtin_nontemporal_load(ph + o - B * H) : 0.f;
                float dht = dh[r * HS + c];
                if (pdy) dht += __builtin_nontemporal_load(pdy + o);
                if (pdl && last) dht += pdl[(size_t)gc * H + c];
                if (!ok) dht = 0.f;
                const float dn = dht * (1.0f - zz);
                const float dz = dht * (hp - nn);
                const float dhc = dht * zz;
                const float dan = dn * (1.0f - nn * nn);
                const float danr = dan * rr;
                const float dar = dan * gg * rr * (1.0f - rr);
                const float daz = dz * zz * (1.0f - zz);
                if (ok) {
                    const size_t og = ((size_t)t * B + g) * (4 * H) + c;
                    __builtin_nontemporal_store(dar, og4 + og); __builtin_nontemporal_store(daz, og4 + og + H);
                    __builtin_nontemporal_store(dan, og4 + og + 2 * H); __builtin_nontemporal_store(danr, og4 + og + 3 * H);
                }
                dG[r * GS + c] = dar; dG[r * GS + H + c] = daz; dG[r * GS + 2 * H + c] = dan; dG[r * GS + 3 * H + c] = danr;
                dh[r * HS + c] = dhc;            // the z * dh part of dh_{t-1}; the matrix part is added below
            }
        }
        }
        osg::lds_barrier();
        if (PF && t > 0) prefetch(t - 1);
        // ---- dx_t and dh_{t-1} (MFMA), output chunks dealt round-robin to the four waves ----
        f32x16 deferred[RB];
        int deferred_oc = -1;
        auto dh_add = [&](int oc, const f32x16 (&v)[RB]) {
            float *dhp = dh + (4 * lh) * HS + oc * 32 + li;
            float old[RB][16];
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int e = 0; e < 16; e++) old[rb][e] = dhp[(rb * 32 + (e & 3) + 8 * (e >> 2)) * HS];
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int e = 0; e < 16; e++) dhp[(rb * 32 + (e & 3) + 8 * (e >> 2)) * HS] = old[rb][e] + v[rb][e];
        };
        for (int item = wave; item < (nchx + nchh) * qsplit; item += NW) {
            const int ch = item / qsplit, qh = item % qsplit;
            const bool is_h = ch >= nchx;
            const int oc = is_h ? ch - nchx : ch;
            const float *wp = (is_h ? a.whhT : a.wihT) + (size_t)oc * Q * 64;
            const int qlo = qh * (Q / qsplit), qhi = qsplit == 1 ? Q : (qh + 1) * (Q / qsplit);   // Q = 3H/2 is even
            f32x16 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rb][e] = 0.f;
            // D-deep software pipeline in blocks of D k-pairs: B fragments by buffer loads (wave-uniform descriptor, SGPR block
            // offset, the position inside the block in the instruction's immediate field), A fragments from the LDS tile with a
            // per-block base pointer and immediate offsets; the n-part of the recurrent path reads the r-scaled section of the
            // tile (columns >= 2H shift by H: H is a multiple of D, so a block never straddles that boundary).  No per-k-pair
            // address arithmetic or bounds tests: the reduction length is a multiple of D by construction (PMC before: 19 VALU
            // and 36 SALU instructions per MFMA in this loop, on the pipe the fp32 MFMA shares).
            const osk::rsrc_t rw = osk::make_rsrc(wp, (uint32_t)Q * 256u);
            const uint32_t wl = (uint32_t)lane * 4u;
            const float *arow[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) arow[rb] = dG + (rb * 32 + li) * GS + lh;
            // resident part: k-pairs qlo .. qlo + WR - 1 against the register-held weights, A fragments one block of 16 ahead
            const int qs = qlo + WR;                           // first streamed k-pair
            auto resident = [&]() {
                if constexpr (WR > 0) {
                    constexpr int DA = 16;
                    float ab[DA];
                    const int sh0 = __builtin_amdgcn_readfirstlane(2 * qlo + ((is_h && qlo >= H) ? H : 0));
#pragma unroll
                    for (int d = 0; d < DA; d++) ab[d] = arow[0][sh0 + 2 * d];
#pragma unroll
                    for (int b = 0; b < WR / DA; b++) {
                        const int qn = qlo + (b + 1) * DA;
                        const int sh = __builtin_amdgcn_readfirstlane(2 * qn + ((is_h && qn >= H) ? H : 0));
#pragma unroll
                        for (int d = 0; d < DA; d++) {
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[d], wreg[b * DA + d], acc[0], 0, 0, 0);
                            if (b + 1 < WR / DA) ab[d] = arow[0][sh + 2 * d];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            auto run = [&](auto dc) {
                constexpr int D = decltype(dc)::value;
                float wbf[D], abf[D][RB];
                auto fetch = [&](int qb) {                     // k-pairs qb .. qb + D - 1 into the ring
                    const uint32_t wo = __builtin_amdgcn_readfirstlane((uint32_t)qb * 256u);
                    const int sh = __builtin_amdgcn_readfirstlane(2 * qb + ((is_h && qb >= H) ? H : 0));
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, wo);
#pragma unroll
                        for (int rb = 0; rb < RB; rb++) abf[d][rb] = arow[rb][sh + 2 * d];
                    }
                };
                fetch(qs);
                resident();                                    // needs nothing from memory: the loads above fly underneath
                for (int q0 = qs; q0 + D < qhi; q0 += D) {
                    const int qn = q0 + D;
                    const uint32_t wo = __builtin_amdgcn_readfirstlane((uint32_t)qn * 256u);
                    const int sh = __builtin_amdgcn_readfirstlane(2 * qn + ((is_h && qn >= H) ? H : 0));
#pragma unroll
                    for (int d = 0; d < D; d++) {
#pragma unroll
                        for (int rb = 0; rb < RB; rb++)
                            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], acc[rb], 0, 0, 0);
                        wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, wo);
#pragma unroll
                        for (int rb = 0; rb < RB; rb++) abf[d][rb] = arow[rb][sh + 2 * d];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int d = 0; d < D; d++)
#pragma unroll
                    for (int rb = 0; rb < RB; rb++)
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], acc[rb], 0, 0, 0);
            };
            // one MFMA per k-pair and row block: 16 x 64 cycles cover an L2 round trip
            if (qs >= qhi) resident();
            else if (((qhi - qs) & 15) == 0 && RB == 1) run(std::integral_constant<int, 16>{});
            else run(std::integral_constant<int, 8>{});
            const osk::rsrc_t rdx = osk::make_rsrc(a.dx, (uint32_t)((size_t)a.T * B * K * 4));
            const uint32_t dxl = (uint32_t)((4 * lh) * K + oc * 32 + li) * 4u;
            // dh += acc as sixteen LDS reads in flight, then sixteen writes.  (Element-wise read - wait - add - write behind
            // four branches each cost 3 us of every step; ds_add_f32 is worse still -- LDS float atomics ran the whole launch
            // 0.06 ms slower.)  The second half of a split chunk waits for the first behind a barrier, below the loop.
            if (is_h && qh == 1) {
#pragma unroll
                for (int rb = 0; rb < RB; rb++) deferred[rb] = acc[rb];
                deferred_oc = oc;
            } else if (is_h) {
                dh_add(oc, acc);
            } else {
                const bool cok = oc * 32 + li < K;
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int rr = rb * 32 + (e & 3) + 8 * (e >> 2);
                        if (cok && row0 + rr + 4 * lh < a.B) {
                            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(((size_t)t * B + row0 + rr) * K * 4));
                            const float v = acc[rb][e];        // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0, hipcc 7.0)
                            if (STACK) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rdx, dxl, so, 17);
                            else osk::buf_store(rdx, dxl, so, v);
                        }
                    }
            }
        }
        if (qsplit == 2) {                                     // a wave has at most one item when chunks are split
            osg::lds_barrier();
            if (deferred_oc >= 0) dh_add(deferred_oc, deferred);
        }
        if (STACK && a.flag_mine && a.need_dx) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's dx stores are acknowledged
        osg::lds_barrier();
        if constexpr (STACK) {
            if (a.flag_mine && a.need_dx && threadIdx.x == 0 && !(a.drop_from >= 0 && a.T - 1 - t >= a.drop_from))
                __hip_atomic_store(a.flag_mine, (uint32_t)(a.T - t), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int RB, int NW, int WR = 0>
__global__ __launch_bounds__(NW * 64, 1) void bwd_sweep_kernel(const SweepArgs a)
{
    sweep_body<RB, NW, WR, false>(a);
}

// ---- bwd_sweep_wide_kernel: the stacked sweep on FOUR compute units per (layer, tile) (the backward twin of gru_wide_kernel.hip) ----
// bwd_sweep_stack_kernel runs a (layer, tile) on one CU: 8 - 10 output chunks x 192 k-pairs = 12.7 us of fp32 MFMA per step there,
// 18 us per pipeline stage measured at the reference's batch of 64.  Here group q of (layer, tile) is its own workgroup and produces
// output chunk q of dh_{t-1} (and, above the bottom layer, chunk q of dx_t): its waves split that chunk's 192 k-pair reduction 4 (8)
// ways and keep their 48 (24) weight fragments in registers for the whole launch.  Every group forms the gate derivatives of the WHOLE
// tile (cheap VALU work on prefetched activations) and stores its own quarter of them for the dW kernels; per step the four groups
// exchange their 32-column slices of dh_{t-1} through a [T][B][H] buffer, and hand their slices of dx_t to the layer below, with
// write-through stores and one counter per group (steps published = T - t), polled with cache-bypassing loads.  Bounded waits, error
// word, NaN poisoning and the -20 return as in the other progress-counter kernels.  Block index -> (tile, layer, group) keeps a tile's
// workgroups on one XCD.  Same sums as bwd_sweep_kernel up to the order of the partial reductions.
typedef float f32x4w __attribute__((ext_vector_type(4)));
struct SweepWideArgs {
    int n, tiles;
    uint32_t *flags;             // [n][tiles][4], zeroed before the launch
    int32_t *err, *err_local;
    uint32_t max_polls;
    int drop_y, drop_step;
    float *dhx[8];               // per launch row: dh exchange buffer [T][B][H] (index t holds the gradient w.r.t. h_t from the later steps)
    SweepArgs layer[8];          // layer[0] = the top layer
};

__global__ __launch_bounds__(512, 1) void bwd_sweep_wide_kernel(const SweepWideArgs sa)
{
    constexpr int H = 128, HS = H + 1, GS = 4 * H + 1, Q = 3 * H / 2, NKW = 48;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // dh [32][129] | dG [32][513] | partial sums [8 waves][4 quads][64][4]
    float *dh = sm, *dG = sm + 32 * HS, *xch = dG + 32 * GS;        // (32 * (129 + 513) = 20,544 floats: the exchange area is 16-byte aligned)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = 4 * sa.n;
    const int tile = (slot / per) * 8 + xcd, y = (slot % per) >> 2, q = slot & 3;
    if (tile >= sa.tiles) return;
    const SweepArgs &a = sa.layer[y];
    const int B = a.B, T = a.T, K = a.K, row0 = tile * 32;
    // work items of this group: output chunk q of W_hh^T (item 0) and, with need_dx, chunk q of W_ih^T (item 1: K = 128 above the
    // bottom layer); a chunk's 192 k-pairs over 8 / items waves
    const int items = a.need_dx ? 2 : 1, slices = 8 / items, nk = Q / slices;
    const int item = items == 2 ? wave >> 2 : 0, sl = items == 2 ? wave & 3 : wave, qlo = sl * nk;
    const bool is_h = item == 0;
    float wreg[NKW];
    {
        const float *wp = (is_h ? a.whhT : a.wihT) + (size_t)q * Q * 64 + (size_t)qlo * 64;
#pragma unroll
        for (int d = 0; d < NKW; d++) wreg[d] = d < nk ? wp[d * 64 + lane] : 0.f;
    }
    const osk::rsrc_t rf = osk::make_rsrc(sa.flags, (uint32_t)sa.n * (uint32_t)sa.tiles * 16u);
    const uint32_t own_off = (uint32_t)((y * sa.tiles + tile) * 16), up_off = y > 0 ? (uint32_t)(((y - 1) * sa.tiles + tile) * 16) : 0u;
    bool lost = osg::stack_lost_already(sa.err_local);
    for (int i = threadIdx.x; i < 32 * HS; i += 512) dh[i] = 0.f;
    __syncthreads();

    // element map of the VALU phase: thread -> column c = tid % 128, rows tid / 128 + 4 e
    const int c = threadIdx.x & (H - 1), rbase = threadIdx.x >> 7;
    uint32_t eoff[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int g = row0 + rbase + 4 * e;
        eoff[e] = (uint32_t)(((size_t)(g < B ? g : B - 1) * H + c) * 4);
    }
    const uint32_t step_bytes = (uint32_t)B * H * 4u, all_bytes = (uint32_t)((size_t)T * B * H * 4);
    float pf_r[8], pf_z[8], pf_n[8], pf_g[8], pf_h[8];
    auto prefetch = [&](int t) {
        const uint32_t st = (uint32_t)((size_t)t * B * H * 4), hst = t > 0 ? st - step_bytes : 0u;
        const osk::rsrc_t rr_ = osk::make_rsrc(a.sv_r, all_bytes), rz_ = osk::make_rsrc(a.sv_z, all_bytes), rn_ = osk::make_rsrc(a.sv_n, all_bytes),
                          rg_ = osk::make_rsrc(a.sv_g, all_bytes), rh_ = osk::make_rsrc(a.sv_h, all_bytes);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            pf_r[e] = osk::buf_load_nt(rr_, eoff[e], st); pf_z[e] = osk::buf_load_nt(rz_, eoff[e], st); pf_n[e] = osk::buf_load_nt(rn_, eoff[e], st);
            pf_g[e] = osk::buf_load_nt(rg_, eoff[e], st); pf_h[e] = osk::buf_load_nt(rh_, eoff[e], hst);
        }
    };
    prefetch(T - 1);

    OSL_TS_DECL
    for (int t = T - 1; t >= 0; t--) {
        OSL_TS(0)
        // ---- dh_t (the four groups' slices from step t + 1) and dy_t (the layer above, its step t): one wait, all loads together ----
        const uint32_t need_own = (uint32_t)(T - 1 - t), need_up = y > 0 ? (uint32_t)(T - t) : 0u;
        if (!lost && (need_own || need_up)) {
            const uint32_t need = lane < 4 ? need_own : need_up;
            const uint32_t off = (lane < 4 ? own_off : up_off) + (uint32_t)(lane & 3) * 4u;
            lost = true;
            for (uint32_t spin = 0; spin < sa.max_polls; spin++) {
                uint32_t v = need;
                if (lane < 8 && need) v = __builtin_amdgcn_raw_buffer_load_b32(rf, off, 0u, 17);
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) { lost = false; break; }
                if ((spin & 1023u) == 1023u && osg::stack_lost_already(sa.err_local)) break;      // somebody else gave up: so do we
                __builtin_amdgcn_s_sleep(1);
            }
            if (lost && lane == 0) {
                if (sa.err_local) __hip_atomic_fetch_or(sa.err_local, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sa.err) __hip_atomic_fetch_or(sa.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        float vd[8], vy[8];
        {
            const osk::rsrc_t rd = osk::make_rsrc(sa.dhx[y] + (size_t)t * B * H, step_bytes);
            const osk::rsrc_t ry = osk::make_rsrc((a.dy ? a.dy : a.sv_r) + (size_t)t * B * H, step_bytes);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                vd[e] = t < T - 1 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, eoff[e], 0u, 17)) : 0.f;
                vy[e] = y > 0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, eoff[e], 0u, 17)) : 0.f;
            }
        }
        OSL_TS(1)                                               // counters + dh / dy tiles requested
        // ---- gate derivatives of the whole tile (every group), this group's quarter of them to memory for the dW kernels ----
        {
            const bool last = t == T - 1;
            const osk::rsrc_t rg4 = osk::make_rsrc(a.dg4 + (size_t)t * B * 4 * H, (uint32_t)B * 4u * H * 4u);
            const bool mine = (c >> 5) == q;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int r = rbase + 4 * e, g = row0 + r;
                const bool ok = g < B;
                const float rr = pf_r[e], zz = pf_z[e], nn = pf_n[e], gg = pf_g[e], hp = t > 0 ? pf_h[e] : 0.f;
                float dht = vd[e] + vy[e];
                if (lost) dht = __builtin_nanf("");
                if (a.dy_last && last) dht += a.dy_last[(size_t)(ok ? g : B - 1) * H + c];
                if (!ok) dht = 0.f;
                const float dn = dht * (1.0f - zz);
                const float dz = dht * (hp - nn);
                const float dhc = dht * zz;
                const float dan = dn * (1.0f - nn * nn);
                const float danr = dan * rr;
                const float dar = dan * gg * rr * (1.0f - rr);
                const float daz = dz * zz * (1.0f - zz);
                if (ok && mine) {
                    const uint32_t og = (uint32_t)(((size_t)g * 4 * H + c) * 4);
                    osk::buf_store_nt(rg4, og, 0u, dar); osk::buf_store_nt(rg4, og, (uint32_t)H * 4u, daz);
                    osk::buf_store_nt(rg4, og, 2u * H * 4u, dan); osk::buf_store_nt(rg4, og, 3u * H * 4u, danr);
                }
                dG[r * GS + c] = dar; dG[r * GS + H + c] = daz; dG[r * GS + 2 * H + c] = dan; dG[r * GS + 3 * H + c] = danr;
                dh[r * HS + c] = dhc;            // the z * dh part of dh_{t-1}; the matrix part is added at the reduction
            }
        }
        __syncthreads();
        OSL_TS(2)                                               // tiles landed, gate derivatives, barrier
        if (t > 0) prefetch(t - 1);
        // ---- this wave's slice of its chunk's reduction: W_hh^T reads sections da_r, da_z, da_n r (columns >= 2H shift by H), W_ih^T 0..3H;
        // then the partial sums through LDS as 16-byte pieces: wave `sl` of an item sums elements [sl epw, (sl + 1) epw) over the item's
        // waves and stores them.  Compile-time in the item count: the A fragments of the next eight k-pairs are requested before the
        // MFMAs of the current eight, and the reduction's LDS reads are all in flight together (with run-time bounds this phase took
        // 6 k cycles of a step's 18 k) ----
        auto mfma_reduce = [&](auto items_c) {
            constexpr int ITEMS = decltype(items_c)::value, SLICES = 8 / ITEMS, NK = Q / SLICES, EPW = 16 / SLICES;      // 48 / 24 k-pairs, 4 / 2 elements
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            const float *arow = dG + li * GS + lh;
            float av[2][8];
            auto fetch = [&](int d0, float *dst) {
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    const int kk = qlo + d0 + d;
                    dst[d] = arow[2 * kk + ((is_h && kk >= H) ? H : 0)];
                }
            };
            fetch(0, av[0]);
#pragma unroll
            for (int b = 0; b < NK / 8; b++) {
                if (b + 1 < NK / 8) fetch(8 * (b + 1), av[(b + 1) & 1]);
#pragma unroll
                for (int d = 0; d < 8; d++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][d], wreg[8 * b + d], acc, 0, 0, 0);
            }
            OSL_TS(3)                                           // MFMA slice
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                *reinterpret_cast<f32x4w *>(xch + (size_t)((wave * 4 + qd) * 64 + lane) * 4) = (f32x4w){acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
            __syncthreads();
            float own[EPW];
#pragma unroll
            for (int k = 0; k < EPW; k++) own[k] = 0.f;
            if constexpr (EPW == 4) {
#pragma unroll
                for (int j = 0; j < SLICES; j++) {
                    const f32x4w v = *reinterpret_cast<const f32x4w *>(xch + (size_t)(((item * SLICES + j) * 4 + sl) * 64 + lane) * 4);
                    own[0] += v[0]; own[1] += v[1]; own[2] += v[2]; own[3] += v[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < SLICES; j++) {
                    const osk::f2 v = *reinterpret_cast<const osk::f2 *>(xch + (size_t)((j * 4 + (sl >> 1)) * 64 + lane) * 4 + 2 * (sl & 1));
                    own[0] += v[0]; own[1] += v[1];
                }
            }
            const osk::rsrc_t rdh = osk::make_rsrc(sa.dhx[y] + (size_t)(t > 0 ? t - 1 : 0) * B * H, t > 0 ? step_bytes : 0u);      // (t = 0: nobody reads dh_{-1})
            const osk::rsrc_t rdx = osk::make_rsrc(a.dx + (size_t)t * B * K, (uint32_t)B * (uint32_t)K * 4u);
#pragma unroll
            for (int k = 0; k < EPW; k++) {
                // element e = sl EPW + k -> row (e & 3) + 8 (e >> 2) + 4 lh
                const int row = EPW == 4 ? k + 8 * sl + 4 * lh : (2 * (sl & 1) + k) + 8 * (sl >> 1) + 4 * lh;
                if (is_h) {
                    const float v = own[k] + dh[row * HS + q * 32 + li];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rdh, (uint32_t)(((size_t)(row0 + row) * H + q * 32 + li) * 4), 0u, 17);
                } else {
                    const float v = own[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rdx, (uint32_t)(((size_t)(row0 + row) * K + q * 32 + li) * 4), 0u, 17);
                }
            }
        };
        if (items == 2) mfma_reduce(std::integral_constant<int, 2>{});
        else mfma_reduce(std::integral_constant<int, 1>{});
        OSL_TS(4)                                               // partial sums, reduction, slice stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slice stores (and its gate-derivative stores) are acknowledged
        __syncthreads();
        if (threadIdx.x == 0 && !(y == sa.drop_y && T - 1 - t >= sa.drop_step && sa.drop_step >= 0))
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(T - t), rf, own_off + (uint32_t)q * 4u, 0u, 17);
        OSL_TS(5)                                               // acknowledgements (incl. the next step's prefetch), barrier, counter
    }
#ifdef OS_LAYER_TS
    if (tile == 0 && q == 0 && threadIdx.x == 0)
        printf("bwd_sweep_wide_kernel row %d (0 = top) T=%d cycles per step (wave 0): wait + tile requests %llu | tiles + gate derivatives + barrier %llu | MFMA slice %llu | partials + reduction + stores %llu | acks + barrier + counter %llu | sum %llu\n",
               y, T, ts_sum[1] / T, ts_sum[2] / T, ts_sum[3] / T, ts_sum[4] / T, ts_sum[5] / T, (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5]) / T);
#endif
}

struct SweepStackArgs {
    int n, tiles, y0;            // y0: first layer index of this launch (0; the debug sequence OS_SWEEP_STACK_DBG=1 launches one layer at a time)
    uint32_t *flags;             // [n][tiles], zeroed before the launch
    int32_t *err, *err_local;    // the context's error word and its device twin
    uint32_t max_polls;
    int drop_y, drop_step;       // tests: launch row drop_y (0 = the top layer) stops publishing after drop_step steps (-1: never)
    SweepArgs layer[8];          // layer[0] = the top layer
};
#ifndef OS_SWEEP_STACK_WR
#define OS_SWEEP_STACK_WR 16     // leading k-pairs of a wave's weight chunk kept in registers: 32 spills two VGPRs here (12 B of scratch)
#endif
__global__ __launch_bounds__(512, 1) void bwd_sweep_stack_kernel(const SweepStackArgs sa)
{
    const int y = blockIdx.y + sa.y0;
    SweepArgs a = sa.layer[y];
    a.flag_mine = sa.flags + (size_t)y * sa.tiles + blockIdx.x;
    a.flag_prev = y > 0 ? a.flag_mine - sa.tiles : nullptr;
    a.err = sa.err; a.err_local = sa.err_local; a.max_polls = sa.max_polls; a.drop_from = y == sa.drop_y ? sa.drop_step : -1;
    sweep_body<1, 8, OS_SWEEP_STACK_WR, true>(a);
}

// ---- weight gradients: dW[3H][K] += dG^T X over a slice of the T*B rows, db[3H] += column sums of dG ----
// Both MFMA operands are coalesced straight from HBM/L2: for a 2-row step, lane (i = l&31, kk = l>>5) supplies
// A = dG[r+kk][j0+i] (32 consecutive gate units) and B = X[r+kk][k0+i] (32 consecutive inputs); a wave keeps one 32-wide
// gate chunk and up to 6 input chunks (96 accumulator registers), so dG is read once per wave and X once per gate chunk
// (it stays in the 256 MiB Infinity Cache).  The T*B-long reduction is split over blockIdx.y; partial tiles are added
// with fp32 atomics (two 128-B row segments per wave instruction).  The bias gradient rides along on the VALU.
struct DwArgs {
    int H3, K;                // gate units, input width
    size_t r_begin, r_end;    // rows of dG to reduce (row r of dG pairs with row r - x_row_shift of X)
    size_t x_row_shift;
    size_t x_valid_from;      // dw2_kernel: rows below this pair with X = 0 (h_{-1}): they only count for the bias sum (multiple of 32)
    int rows_per_slice;
    const float *dG;          // [rows][ldg]: gate unit j sits in column j (+ nshift for the n gate, j >= 2 H3 / 3)
    int ldg, nshift;          // the sweep's four-section array: ldg = 4H; nshift = 0 for W_ih (da_n), H for W_hh (da_n * r)
    const float *X;           // [rows][K], or (B, T, K) batch_first when x_btf (row r = t*B + b lives at (b*T + t)*K)
    int x_btf, B, T;
    float *dW;                // [H3][K]
    float *db;                // [H3] or null
};

// Workgroup = 4 waves = 4 gate chunks (wave w owns chunk 4*blockIdx.x + w) over one slice of rows.  The X rows of a
// 32-row tile are staged ONCE per workgroup through LDS (register-staged double buffering: the global loads of tile i+1
// are in flight while tile i feeds 16 x 6 MFMAs per wave), so X is read from L2/HBM once per 4 gate chunks instead of once
// per chunk; each wave streams its own dG column chunk straight from memory, one step ahead.
constexpr int DW_TR = 32;            // rows per tile

// two workgroups per CU for the float4 variant (the scalar-staging fallback needs the full register file).
// NC = 32-column input chunks per pass (accumulators per wave): 6 for the 188-wide first layer, 4 for the 128-wide ones --
// with a fixed 6 the 128-wide products issued half as many MFMAs again on zero columns (PMC: 2.8 M matrix instructions per
// launch against 2.0 M algorithmic).
template <bool VEC4, int NC>
__global__ __launch_bounds__(256, VEC4 ? 2 : 1) void dw_kernel(const DwArgs a)
{
    constexpr int DW_KMAX = NC * 32;     // input columns held per pass
    constexpr int NV4 = DW_TR * DW_KMAX / 4 / 256, NV1 = DW_TR * DW_KMAX / 256;      // staging float4 / floats per thread
    __shared__ __attribute__((aligned(16))) float Xs[2][DW_TR][DW_KMAX];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, kk = lane >> 5;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    const bool jok = j0 < a.H3;
    const int jc = j0 + (3 * j0 >= 2 * a.H3 ? a.nshift : 0);          // column of the chunk in dG
    const size_t r0 = a.r_begin + (size_t)blockIdx.y * a.rows_per_slice;
    size_t r1 = r0 + a.rows_per_slice;
    if (r1 > a.r_end) r1 = a.r_end;
    if (r0 >= r1) return;
    const int nkc = (a.K + 31) / 32;
    const int ntiles = (int)((r1 - r0 + DW_TR - 1) / DW_TR);
    float bsum = 0.f;
    auto xrow_off = [&](size_t rr) -> size_t {
        const uint32_t xr = (uint32_t)(rr - a.x_row_shift);            // row counts are far below 2^32: 32-bit div/mod
        return a.x_btf ? ((size_t)(xr % (uint32_t)a.B) * a.T + xr / (uint32_t)a.B) * (size_t)a.K : (size_t)xr * (size_t)a.K;
    };
    for (int kc0 = 0; kc0 < nkc; kc0 += NC) {
        const int kbase = kc0 * 32, kw = (a.K - kbase) < DW_KMAX ? (a.K - kbase) : DW_KMAX;   // columns of this pass
        f32x16 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[c][e] = 0.f;
        // staging registers: 32 rows x 192 columns / 256 threads = 24 floats per thread.  With K % 4 == 0 (every shape the
        // reference uses: 60, 64, 128, 188) a thread moves six float4 -- six row addresses per tile instead of 24 (the address
        // arithmetic was half of this kernel's VALU time, and VALU time is MFMA time lost: they share the pipe)
        float stg[NV1];
        auto stage_load = [&](int tile) {
            if (VEC4) {
#pragma unroll
                for (int i = 0; i < NV4; i++) {
                    const int f4 = threadIdx.x + 256 * i;                           // 0 .. 1535
                    const int row = f4 / (DW_KMAX / 4), col = 4 * (f4 % (DW_KMAX / 4));
                    const size_t rr = r0 + (size_t)tile * DW_TR + row;
                    const bool ok = tile < ntiles && rr < r1 && col < kw;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) v = *reinterpret_cast<const float4 *>(a.X + xrow_off(rr) + kbase + col);
                    stg[4 * i] = v.x; stg[4 * i + 1] = v.y; stg[4 * i + 2] = v.z; stg[4 * i + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV1; i++) {
                    const int flat = threadIdx.x + 256 * i;                         // 0 .. 6143
                    const int row = flat / DW_KMAX, col = flat % DW_KMAX;
                    const size_t rr = r0 + (size_t)tile * DW_TR + row;
                    const bool ok = tile < ntiles && rr < r1 && col < kw;
                    stg[i] = ok ? a.X[xrow_off(rr) + kbase + col] : 0.f;
                }
            }
        };
        auto stage_store = [&](int buf) {
            if (VEC4) {
#pragma unroll
                for (int i = 0; i < NV4; i++) {
                    const int f4 = threadIdx.x + 256 * i;
                    *reinterpret_cast<float4 *>(&Xs[buf][f4 / (DW_KMAX / 4)][4 * (f4 % (DW_KMAX / 4))]) =
                        make_float4(stg[4 * i], stg[4 * i + 1], stg[4 * i + 2], stg[4 * i + 3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV1; i++) {
                    const int flat = threadIdx.x + 256 * i;
                    Xs[buf][flat / DW_KMAX][flat % DW_KMAX] = stg[i];
                }
            }
        };
        stage_load(0);
        stage_store(0);
        __syncthreads();
        // this wave's dG operand: the 16 two-row steps of a tile are fetched one whole tile ahead (register double buffer), so a
        // full tile of MFMAs (16 x 6 x 64 cycles) covers the HBM round trip instead of one step's worth
        float avb[2][DW_TR / 2];
        auto dg_load = [&](int tile, float *dst) {
            const size_t rt = r0 + (size_t)tile * DW_TR;
#pragma unroll
            for (int st = 0; st < DW_TR / 2; st++) {
                const size_t rr = rt + 2 * st + kk;
                dst[st] = (jok && tile < ntiles && rr < r1) ? __builtin_nontemporal_load(a.dG + rr * a.ldg + jc + li) : 0.f;
            }
        };
        dg_load(0, avb[0]);
        for (int tile = 0; tile < ntiles; tile += 2) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int tl = tile + half, buf = half;            // ntiles may be odd: the last half-iteration is skipped
                if (tl < ntiles) {
                    stage_load(tl + 1);                             // X tile: in flight during this tile's MFMAs
                    dg_load(tl + 1, avb[half ^ 1]);
#pragma unroll
                    for (int st = 0; st < DW_TR / 2; st++) {
                        const float av = avb[half][st];
                        if (kc0 == 0) bsum += av;
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Xs[buf][2 * st + kk][c * 32 + li], acc[c], 0, 0, 0);
                        if (st & 1) __builtin_amdgcn_sched_barrier(0);      // keep the 96 LDS reads of a tile from being hoisted (spills)
                    }
                    stage_store(buf ^ 1);
                }
                __syncthreads();
            }
        }
        if (jok) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int k = kbase + c * 32 + li;
                if (k < a.K) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                        atomicAdd(&a.dW[(size_t)j * a.K + k], acc[c][e]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (a.db && jok) {
        bsum += __shfl_xor(bsum, 32, 64);            // the two row parities of the same gate unit
        if (kk == 0) global_fadd(a.db, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsum);
    }
}

// ---- dw2_kernel: the same reduction with (almost) no address arithmetic in the loop (K % 4 == 0) ----
// PMC on dw_kernel at the training batch: 7.4 VALU + 3.4 SALU instructions per MFMA (register staging of the X tile with 64-bit
// addresses, float4 repacking, double-buffer copies), on the pipe the fp32 MFMA shares.  Here:
//   * the X tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write); a thread's
//     16-byte pieces sit at fixed (row, column) positions of the tile, computed once; the LDS row pitch is the compile-time
//     NC * 32 floats, so every B-fragment read is one base register + an immediate offset;
//   * the dG fragments come through a buffer descriptor over the slice (rows past its end read as zero by the hardware's
//     range check, which also neutralises whatever the tail tile holds), offset = constant per lane + SGPR;
//   * the partial tiles are added with buffer atomics (SGPR row offsets).
// One barrier per 32-row tile; tile i+1's DMA and dG loads fly underneath tile i's 16 x NC MFMAs per wave.
// (non-template helpers: inside the kernel template hipcc's host pass silently drops the kernel's launch stub when it meets
// these builtins)
__device__ __forceinline__ void lds_dma16(const float *g, float *l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
// buffer form: range-checked, the tile's row base in an SGPR offset, the piece's position inside the tile in one VGPR
__device__ __forceinline__ void lds_dma16_buf(osk::rsrc_t r, float *l, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void buf_atomic_add(float v, osk::rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, voff, soff, 0);
}

// (Tried and not kept: ONE 768-thread workgroup per CU holding all twelve gate chunks of H = 128, X staged once, cu_count
// equal slices: 0.226 ms per layer against 0.196 -- twelve waves behind one barrier per tile lose more than the even split wins.)
template <int NC>
__global__ __launch_bounds__(256, 2) void dw2_kernel(const DwArgs a)
{
    constexpr int NWV = 4;
    constexpr int PITCH = NC * 32;                       // floats per LDS row
    constexpr int PPR = PITCH / 4;                       // 16-byte pieces per LDS row
    constexpr int NTH = NWV == 12 ? 512 : 256;           // threads that stage X (the first eight waves of the 12-wave form)
    constexpr int NP = DW_TR * PPR / NTH;                // pieces per staging thread per tile
    __shared__ __attribute__((aligned(16))) float Xs[2][DW_TR * PITCH + 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, kk = lane >> 5;
    const int j0 = (blockIdx.x * NWV + wave) * 32;
    const bool jok = j0 < a.H3;
    const size_t r0 = a.r_begin + (size_t)blockIdx.y * a.rows_per_slice;
    size_t r1 = r0 + a.rows_per_slice;
    if (r1 > a.r_end) r1 = a.r_end;
    if (r0 >= r1) return;
    const int nkc = (a.K + 31) / 32;
    const int ntiles = (int)((r1 - r0 + DW_TR - 1) / DW_TR);
    const uint32_t nrows = (uint32_t)(r1 - r0);
    // this wave's dG column chunk of the slice: [nrows][H3] starting at row r0; a wave without a chunk reads zeros
    const int jc = j0 + (3 * j0 >= 2 * a.H3 ? a.nshift : 0);          // column of the chunk in dG
    const osk::rsrc_t rg = osk::make_rsrc(a.dG + r0 * a.ldg, jok ? nrows * (uint32_t)a.ldg * 4u : 0u);
    const uint32_t gl = (uint32_t)(kk * a.ldg + jc + li) * 4u, grow = (uint32_t)a.ldg * 8u;     // two rows per step
    float bsum = 0.f;
    for (int kc0 = 0; kc0 < nkc; kc0 += NC) {
        const int kbase = kc0 * 32, kw = (a.K - kbase) < PITCH ? (a.K - kbase) : PITCH;       // columns of this pass
        // Fixed position of this thread's pieces inside a tile: ONE byte offset each (rows of a tile are `rstride` floats
        // apart: K, or T*K for the batch_first input, where a 32-row tile is 32 consecutive windows of one time step -- the
        // host only picks this kernel for that layout when B and the slices are multiples of 32).  The tile's row base rides in
        // the SGPR offset and the range check zero-fills what lies past X, so a tile costs a few scalar instructions instead of
        // ~50 vector ones per piece (64-bit clamps, div/mod for the batch_first rows): measured 19 k of a workgroup's 105 k
        // cycles, on the pipe the fp32 MFMAs need.  A piece in the padding columns of a pass points out of range (zeros).
        const uint32_t rstride = a.x_btf ? (uint32_t)a.T * (uint32_t)a.K : (uint32_t)a.K;
        uint32_t pvo[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = (threadIdx.x & (NTH - 1)) + NTH * i;
            const int prow = p / PPR, c4 = 4 * (p % PPR);
            pvo[i] = c4 < kw ? ((uint32_t)prow * rstride + (uint32_t)(kbase + c4)) * 4u : 0x80000000u;
        }
        const osk::rsrc_t rxs = osk::make_rsrc(a.X, (uint32_t)((size_t)a.T * a.B * a.K * 4));
        auto dma = [&](int tile, int buf) {
            if (NWV == 12 && wave >= 8) return;          // wave-uniform
            const uint32_t xr0 = (uint32_t)(r0 - a.x_row_shift) + (uint32_t)tile * DW_TR;          // first X row of the tile
            const uint32_t so = __builtin_amdgcn_readfirstlane(
                (a.x_btf ? (xr0 % (uint32_t)a.B) * (uint32_t)a.T + xr0 / (uint32_t)a.B : xr0) * (uint32_t)a.K * 4u);
#pragma unroll
            for (int i = 0; i < NP; i++)
                // LDS destination: wave-uniform base + lane * 16 (the pieces of one instruction are contiguous in the tile image)
                lds_dma16_buf(rxs, &Xs[buf][(wave * 64 + NTH * i) * 4], pvo[i], so);
        };
        f32x16 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[c][e] = 0.f;
        float avb[2][DW_TR / 2];
        auto dg_load = [&](int tile, float *dst) {
            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)tile * (uint32_t)(DW_TR / 2) * grow);
#pragma unroll
            for (int st = 0; st < DW_TR / 2; st++) dst[st] = osk::buf_load_nt(rg, gl, so + (uint32_t)st * grow);
        };
        const float bw = (kc0 == 0 && a.db) ? 1.0f : 0.0f;          // bias gradient rides along in the first pass
        // tiles whose rows pair with h_{-1} = 0 (first time step of the recurrent product) only feed the bias sum
        auto xzero = [&](int tile) { return r0 + (size_t)tile * DW_TR < a.x_valid_from; };
        auto tile_mfma = [&](int buf, const float *av, bool xz) {
            const float *xb = &Xs[buf][kk * PITCH + li];
#pragma unroll
            for (int st = 0; st < DW_TR / 2; st++) bsum = fmaf(av[st], bw, bsum);
            if (xz) return;
            // B fragments one 2-row step ahead of the MFMAs that use them (read - wait - MFMA in lockstep left the LDS latency
            // of every step exposed whenever the other workgroup's wave on the SIMD was not there to cover it)
            float xv[2][NC];
#pragma unroll
            for (int c = 0; c < NC; c++) xv[0][c] = xb[c * 32];
#pragma unroll
            for (int st = 0; st < DW_TR / 2; st++) {
                if (st + 1 < DW_TR / 2) {
#pragma unroll
                    for (int c = 0; c < NC; c++) xv[(st + 1) & 1][c] = xb[2 * (st + 1) * PITCH + c * 32];
                }
#pragma unroll
                for (int c = 0; c < NC; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], xv[st & 1][c], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                  // keep the LDS reads of a tile from being hoisted (spills)
            }
        };
        if (!xzero(0)) dma(0, 0);
        dg_load(0, avb[0]);
        for (int tile = 0; tile < ntiles; tile += 2) {
            __syncthreads();                                        // tile `tile` landed (vmcnt(0) + barrier); buffer 1 is free
            if (tile + 1 < ntiles) { if (!xzero(tile + 1)) dma(tile + 1, 1); dg_load(tile + 1, avb[1]); }
            tile_mfma(0, avb[0], xzero(tile));
            if (tile + 1 < ntiles) {
                __syncthreads();
                if (tile + 2 < ntiles) { if (!xzero(tile + 2)) dma(tile + 2, 0); dg_load(tile + 2, avb[0]); }
                tile_mfma(1, avb[1], xzero(tile + 1));
            }
        }
        if (jok) {
            const osk::rsrc_t rw = osk::make_rsrc(a.dW, (uint32_t)a.H3 * (uint32_t)a.K * 4u);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int k = kbase + c * 32 + li;
                if (k < a.K) {
                    const uint32_t vo = (uint32_t)((4 * kk) * a.K + k) * 4u;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        buf_atomic_add(acc[c][e], rw, vo, (uint32_t)((j0 + (e & 3) + 8 * (e >> 2)) * a.K) * 4u);
                }
            }
        }
        __syncthreads();
    }
    if (a.db && jok) {
        bsum += __shfl_xor(bsum, 32, 64);            // the two row parities of the same gate unit
        if (kk == 0) global_fadd(a.db, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsum);
    }
}

// ---- dw3_kernel: BOTH weight gradients of a layer in one pass over the gate derivatives ------------------------------
// dW_ih += dG_i^T X and dW_hh += dG_h^T H_prev share everything but the n-gate third of dG (da_n against da_n * r) and the
// right-hand matrix, so one launch reads the r and z sections once for both products, stages the X and the h_{t-1} tile of
// a 32-row step side by side, and amortises a tile's barrier, its load issue and the launch itself over twice as many MFMAs
// (NCX + NCH accumulators per wave).  Wave = one 32-wide gate chunk j: A fragment for W_ih from column j, for W_hh from
// column j (+ H in the n gate: two loads per step there, one elsewhere).  Rows of the first time step pair with h_{-1} = 0:
// their tiles skip the recurrent half (whole tiles: B and the slices are multiples of 32).
struct Dw3Args {
    int H3, Kx, H;
    size_t rows;                 // T * B
    int rows_per_slice, ngroups; // ngroups = workgroups (of four gate chunks) per row slice
    const float *dG;             // [rows][ldg]
    int ldg;
    const float *X;              // [rows][Kx], or (B, T, Kx) batch_first when x_btf
    int x_btf, B, T;
    const float *Hp;             // [rows][H]: row r - B pairs with dG row r
    float *dWih, *dWhh, *dbih, *dbhh;
    int dbg;                     // development (OS_DW_DBG): 1 = skip the atomics (timing only), 2 = rotate the epilogue's element order per slice, 4 = stagger odd slices
};

// TR = rows per tile: 32, or 16 for the ten-accumulator form of the 188-wide first layer (160 accumulator registers leave
// room for one 8-step set of gate-derivative fragments per buffer, not for 16)
template <int NCX, int NCH, int TR = DW_TR>
__global__ __launch_bounds__(256, 2) void dw3_kernel(const Dw3Args a)
{
    constexpr int PX = NCX * 32, PH = NCH * 32;             // floats per LDS row
    constexpr int NPX = TR * NCX / 32, NPH = TR * NCH / 32;  // 16-byte pieces per thread per tile
    static_assert(TR * NCX % 32 == 0 && TR * NCH % 32 == 0, "whole pieces per thread");
    __shared__ __attribute__((aligned(16))) float Xs[2][TR * PX + 4];
    __shared__ __attribute__((aligned(16))) float Hs[2][TR * PH + 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, kk = lane >> 5;
    // XCD-aware workgroup order: consecutive workgroup ids go round-robin to the eight XCDs, each with its own L2, and the
    // column groups of one row slice all stream the same X / h tiles -- so ids i, i + 8, i + 16, ... (same XCD, dispatched
    // together) are the column groups of ONE slice, and two of the three re-reads hit that XCD's L2 instead of HBM.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = slot % a.ngroups, by = (slot / a.ngroups) * 8 + xcd;
    const int j0 = (bx * 4 + wave) * 32;
    const bool jok = j0 < a.H3;
    const bool ngate = 3 * j0 >= 2 * a.H3;
    const size_t r0 = (size_t)by * a.rows_per_slice;
    size_t r1 = r0 + a.rows_per_slice;
    if (r1 > a.rows) r1 = a.rows;
    if (r0 >= r1) return;
    const int ntiles = (int)((r1 - r0 + TR - 1) / TR);
    const uint32_t nrows = (uint32_t)(r1 - r0);
    const osk::rsrc_t rg = osk::make_rsrc(a.dG + r0 * a.ldg, jok ? nrows * (uint32_t)a.ldg * 4u : 0u);
    const uint32_t gli = (uint32_t)(kk * a.ldg + j0 + li) * 4u, glh = gli + (ngate ? (uint32_t)a.H * 4u : 0u), grow = (uint32_t)a.ldg * 8u;
    // fixed position of this thread's 16-byte pieces inside the two tiles (see dw2_kernel)
    const uint32_t xstride = a.x_btf ? (uint32_t)a.T * (uint32_t)a.Kx : (uint32_t)a.Kx;
    uint32_t pvx[NPX], pvh[NPH];
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        const int p = threadIdx.x + 256 * i, prow = p / (PX / 4), c4 = 4 * (p % (PX / 4));
        pvx[i] = c4 < a.Kx ? ((uint32_t)prow * xstride + (uint32_t)c4) * 4u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < NPH; i++) {
        const int p = threadIdx.x + 256 * i, prow = p / (PH / 4), c4 = 4 * (p % (PH / 4));
        pvh[i] = ((uint32_t)prow * (uint32_t)a.H + (uint32_t)c4) * 4u;
    }
    const osk::rsrc_t rxs = osk::make_rsrc(a.X, (uint32_t)((size_t)a.T * a.B * a.Kx * 4));
    const osk::rsrc_t rhs = osk::make_rsrc(a.Hp, (uint32_t)((size_t)a.T * a.B * a.H * 4));
    auto hzero = [&](int tile) { return r0 + (size_t)tile * TR < (size_t)a.B; };
    auto dma = [&](int tile, int buf) {
        const uint32_t xr0 = (uint32_t)r0 + (uint32_t)tile * TR;
        const uint32_t sox = __builtin_amdgcn_readfirstlane(
            (a.x_btf ? (xr0 % (uint32_t)a.B) * (uint32_t)a.T + xr0 / (uint32_t)a.B : xr0) * (uint32_t)a.Kx * 4u);
#pragma unroll
        for (int i = 0; i < NPX; i++) lds_dma16_buf(rxs, &Xs[buf][(wave * 64 + 256 * i) * 4], pvx[i], sox);
        if (!hzero(tile)) {
            const uint32_t soh = __builtin_amdgcn_readfirstlane((xr0 - (uint32_t)a.B) * (uint32_t)a.H * 4u);
#pragma unroll
            for (int i = 0; i < NPH; i++) lds_dma16_buf(rhs, &Hs[buf][(wave * 64 + 256 * i) * 4], pvh[i], soh);
        }
    };
    f32x16 accx[NCX], acch[NCH];
#pragma unroll
    for (int c = 0; c < NCX; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) accx[c][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) acch[c][e] = 0.f;
    float avi[2][TR / 2], avh[2][TR / 2];
    auto dg_load = [&](int tile, int buf) {
        const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)tile * (uint32_t)(TR / 2) * grow);
#pragma unroll
        for (int st = 0; st < TR / 2; st++) avi[buf][st] = osk::buf_load_nt(rg, gli, so + (uint32_t)st * grow);
        if (ngate) {
#pragma unroll
            for (int st = 0; st < TR / 2; st++) avh[buf][st] = osk::buf_load_nt(rg, glh, so + (uint32_t)st * grow);
        }
    };
    float bsi = 0.f, bsh = 0.f;
    auto tile_mfma = [&](int buf, bool hz) {
        const float *xb = &Xs[buf][kk * PX + li], *hb = &Hs[buf][kk * PH + li];
#pragma unroll
        for (int st = 0; st < TR / 2; st++) {
            const float ai = avi[buf][st], ah = ngate ? avh[buf][st] : ai;
            bsi += ai; bsh += ah;
            float xv[NCX], hv[NCH];
#pragma unroll
            for (int c = 0; c < NCX; c++) xv[c] = xb[2 * st * PX + c * 32];
            if (!hz) {
#pragma unroll
                for (int c = 0; c < NCH; c++) hv[c] = hb[2 * st * PH + c * 32];
            }
#pragma unroll
            for (int c = 0; c < NCX; c++) accx[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, xv[c], accx[c], 0, 0, 0);
            if (!hz) {
#pragma unroll
                for (int c = 0; c < NCH; c++) acch[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, hv[c], acch[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if ((a.dbg & 4) && (by & 1)) {                              // development: odd slices start half a tile late
        for (int i = 0; i < 32; i++) __builtin_amdgcn_s_sleep(127);
    }
    dma(0, 0);
    dg_load(0, 0);
    for (int tile = 0; tile < ntiles; tile += 2) {
        __syncthreads();                                        // tile `tile` landed (vmcnt(0) + barrier); buffer 1 is free
        if (tile + 1 < ntiles) { dma(tile + 1, 1); dg_load(tile + 1, 1); }
        tile_mfma(0, hzero(tile));
        if (tile + 1 < ntiles) {
            __syncthreads();
            if (tile + 2 < ntiles) { dma(tile + 2, 0); dg_load(tile + 2, 0); }
            tile_mfma(1, hzero(tile + 1));
        }
    }
    if (jok && !(a.dbg & 1)) {
        const osk::rsrc_t rwx = osk::make_rsrc(a.dWih, (uint32_t)a.H3 * (uint32_t)a.Kx * 4u);
        const osk::rsrc_t rwh = osk::make_rsrc(a.dWhh, (uint32_t)a.H3 * (uint32_t)a.H * 4u);
        // The 160 row slices of a column group add into the SAME 48 K addresses, and workgroups that start together reach their
        // epilogues together: the element order is rotated per slice (four orders), so that at any moment the slices in flight
        // hit different cache lines of the gradient instead of queueing on one
        auto epilogue = [&](auto rot) {
            constexpr int R = decltype(rot)::value;
#pragma unroll
            for (int c = 0; c < NCX; c++) {
                const int k = c * 32 + li;
                if (k < a.Kx) {
                    const uint32_t vo = (uint32_t)((4 * kk) * a.Kx + k) * 4u;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int e = (i + R) & 15;
                        buf_atomic_add(accx[c][e], rwx, vo, (uint32_t)((j0 + (e & 3) + 8 * (e >> 2)) * a.Kx) * 4u);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint32_t vo = (uint32_t)((4 * kk) * a.H + c * 32 + li) * 4u;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int e = (i + R) & 15;
                    buf_atomic_add(acch[c][e], rwh, vo, (uint32_t)((j0 + (e & 3) + 8 * (e >> 2)) * a.H) * 4u);
                }
            }
        };
        const int rsel = (a.dbg & 2) ? (by & 3) : 0;
        if (rsel == 0) epilogue(std::integral_constant<int, 0>{});
        else if (rsel == 1) epilogue(std::integral_constant<int, 4>{});
        else if (rsel == 2) epilogue(std::integral_constant<int, 8>{});
        else epilogue(std::integral_constant<int, 12>{});
        bsi += __shfl_xor(bsi, 32, 64);              // the two row parities of the same gate unit
        bsh += __shfl_xor(bsh, 32, 64);
        if (kk == 0) {
            global_fadd(a.dbih, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsi);
            global_fadd(a.dbhh, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsh);
        }
    }
}

// ---- dw3_bf16_kernel<NCX, NCH, SPL> (round 6, OPT-IN: os_gru_set_split_bf16 | OS_GRU_SPLIT_TRAIN; never the default) --------------------
// The same two products (dW_ih += dG_i^T X, dW_hh += dG_h^T H_prev) with every fp32 operand split into SPL bf16 terms and the products
// on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 atomics unchanged): SPL = 3 issues hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid per
// operand pair (everything down to 2^-16 of a product; what is dropped is below fp32 rounding), SPL = 2 hi.hi, hi.lo, lo.hi.  The fp32
// matrix instruction runs at 1/16 of the bf16 rate, so six bf16 products cost 6/16 of one fp32 product's matrix-pipe time.
// A 32-row tile = two 16-row k-blocks.  The X / h_{t-1} tile is split ONCE per workgroup (a thread takes eight consecutive rows of a
// column: the eight k values of one lane's B fragment) and its bf16 terms are written to LDS in fragment order, one ds_read_b128 per
// (chunk, term) and wave afterwards; the gate-derivative column of a wave (A operand) is split in registers.  The tile of step t + 1
// travels in registers underneath the MFMAs of step t.  Same slices, same XCD-aware order, same epilogue as dw3_kernel.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the loads in flight: the next tile's)
__device__ __forceinline__ void lds_barrier_dw()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#ifndef OST_DWBF_OCC
#define OST_DWBF_OCC 1
#endif
template <int NCX, int NCH, int SPL>
__global__ __launch_bounds__(256, OST_DWBF_OCC) void dw3_bf16_kernel(const Dw3Args a)
{
    constexpr int TR = 32, NCT = NCX + NCH, PX = NCX * 32, PH = NCH * 32;
    constexpr int UX = NCX / 2, UH = NCH / 2;                     // (column, 8-row group) units per thread and tile: PX * 4 / 256
    static_assert(NCX % 2 == 0 && NCH % 2 == 0, "whole units per thread");
    // bf16 terms of the tile in B-fragment order: [k-block 2][chunk NCT][term SPL][lane 64] x 16 bytes
    __shared__ __attribute__((aligned(16))) i32x4_t Bt[2 * NCT * SPL * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, kk = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = slot % a.ngroups, by = (slot / a.ngroups) * 8 + xcd;
    const int j0 = (bx * 4 + wave) * 32;
    const bool jok = j0 < a.H3;
    const bool ngate = 3 * j0 >= 2 * a.H3;
    const size_t r0 = (size_t)by * a.rows_per_slice;
    size_t r1 = r0 + a.rows_per_slice;
    if (r1 > a.rows) r1 = a.rows;
    if (r0 >= r1) return;
    const int ntiles = (int)((r1 - r0 + TR - 1) / TR);
    const uint32_t nrows = (uint32_t)(r1 - r0);
    const osk::rsrc_t rg = osk::make_rsrc(a.dG + r0 * a.ldg, jok ? nrows * (uint32_t)a.ldg * 4u : 0u);
    // A operand: rows 16 kb + 8 kk + q of the tile, column j0 + li
    const uint32_t gli = (uint32_t)(8 * kk * a.ldg + j0 + li) * 4u, glh = gli + (ngate ? (uint32_t)a.H * 4u : 0u), grow = (uint32_t)a.ldg * 4u;
    const uint32_t xstride = a.x_btf ? (uint32_t)a.T * (uint32_t)a.Kx : (uint32_t)a.Kx;
    const osk::rsrc_t rxs = osk::make_rsrc(a.X, (uint32_t)((size_t)a.T * a.B * a.Kx * 4));
    const osk::rsrc_t rhs = osk::make_rsrc(a.Hp, (uint32_t)((size_t)a.T * a.B * a.H * 4));
    // this thread's units: u = threadIdx.x + 256 i -> column u % P, row group u / P (k-block rg >> 1, lane half rg & 1)
    uint32_t uxo[UX], uho[UH];
    int uxl[UX], uhl[UH];                                          // LDS slot (in 16-byte units) of term 0
#pragma unroll
    for (int i = 0; i < UX; i++) {
        const int u = threadIdx.x + 256 * i, col = u % PX, rgp = u / PX;
        uxo[i] = col < a.Kx ? ((uint32_t)(8 * rgp) * xstride + (uint32_t)col) * 4u : 0x80000000u;
        uxl[i] = (((rgp >> 1) * NCT + col / 32) * SPL) * 64 + (rgp & 1) * 32 + (col & 31);
    }
#pragma unroll
    for (int i = 0; i < UH; i++) {
        const int u = threadIdx.x + 256 * i, col = u % PH, rgp = u / PH;
        uho[i] = ((uint32_t)(8 * rgp) * (uint32_t)a.H + (uint32_t)col) * 4u;
        uhl[i] = (((rgp >> 1) * NCT + NCX + col / 32) * SPL) * 64 + (rgp & 1) * 32 + (col & 31);
    }
    auto hzero = [&](int tile) { return r0 + (size_t)tile * TR < (size_t)a.B; };
    float vx[UX][8], vh[UH][8];
    auto tile_load = [&](int tile) {
        const uint32_t xr0 = (uint32_t)r0 + (uint32_t)tile * TR;
        const uint32_t sox = __builtin_amdgcn_readfirstlane(
            (a.x_btf ? (xr0 % (uint32_t)a.B) * (uint32_t)a.T + xr0 / (uint32_t)a.B : xr0) * (uint32_t)a.Kx * 4u);
#pragma unroll
        for (int i = 0; i < UX; i++)
#pragma unroll
            for (int q = 0; q < 8; q++) vx[i][q] = osk::buf_load(rxs, uxo[i] + (uint32_t)q * xstride * 4u, sox);
        if (!hzero(tile)) {
            const uint32_t soh = __builtin_amdgcn_readfirstlane((xr0 - (uint32_t)a.B) * (uint32_t)a.H * 4u);
#pragma unroll
            for (int i = 0; i < UH; i++)
#pragma unroll
                for (int q = 0; q < 8; q++) vh[i][q] = osk::buf_load(rhs, uho[i] + (uint32_t)q * (uint32_t)a.H * 4u, soh);
        }
    };
    auto split8 = [&](const float (&v)[8], i32x4_t (&t)[SPL]) {   // eight k values -> SPL fragments of eight bf16 (array references: pointers put the arrays in scratch)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            uint32_t w[SPL];
            osg::split_pair<SPL>(v[2 * jj], v[2 * jj + 1], w);
#pragma unroll
            for (int sp = 0; sp < SPL; sp++) t[sp][jj] = (int)w[sp];
        }
    };
    auto tile_store = [&](bool hz) {
#pragma unroll
        for (int i = 0; i < UX; i++) {
            i32x4_t t[SPL];
            split8(vx[i], t);
#pragma unroll
            for (int sp = 0; sp < SPL; sp++) Bt[uxl[i] + sp * 64] = t[sp];
        }
        if (!hz) {
#pragma unroll
            for (int i = 0; i < UH; i++) {
                i32x4_t t[SPL];
                split8(vh[i], t);
#pragma unroll
                for (int sp = 0; sp < SPL; sp++) Bt[uhl[i] + sp * 64] = t[sp];
            }
        }
    };
    f32x16 accx[NCX], acch[NCH];
#pragma unroll
    for (int c = 0; c < NCX; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) accx[c][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) acch[c][e] = 0.f;
    // the wave's gate-derivative column of tile t + 1 is requested together with the X / h tile, a tile ahead (two register sets)
    float gi0[2][8], gh0[2][8], gi1[2][8], gh1[2][8];         // (separately named sets: a leading buffer dimension put all four in scratch)
    auto dg_load = [&](int tile, float (&gi_)[2][8], float (&gh_)[2][8]) {
        const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)tile * (uint32_t)TR * grow);
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                gi_[kb][q] = osk::buf_load_nt(rg, gli, so + (uint32_t)(16 * kb + q) * grow);
                gh_[kb][q] = ngate ? osk::buf_load_nt(rg, glh, so + (uint32_t)(16 * kb + q) * grow) : 0.f;
            }
    };
    // (weight term, activation term), largest first
    constexpr int NP = SPL == 3 ? 6 : 3;
    constexpr int PW3[6] = {0, 0, 1, 0, 2, 1}, PB3[6] = {0, 1, 0, 2, 0, 1}, PW2[3] = {0, 0, 1}, PB2[3] = {0, 1, 0};
    float bsi = 0.f, bsh = 0.f;
    auto tile_mfma = [&](bool hz, const float (&gi_)[2][8], const float (&gh_)[2][8]) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            i32x4_t Ai[SPL], Ah[SPL];
            split8(gi_[kb], Ai);
            split8(gh_[kb], Ah);
#pragma unroll
            for (int q = 0; q < 8; q++) { bsi += gi_[kb][q]; bsh += ngate ? gh_[kb][q] : gi_[kb][q]; }
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                if (c >= NCX && hz) break;
                i32x4_t Bf[SPL];
#pragma unroll
                for (int sp = 0; sp < SPL; sp++) Bf[sp] = Bt[((kb * NCT + c) * SPL + sp) * 64 + lane];
#pragma unroll
                for (int pi = 0; pi < NP; pi++) {
                    const int wt = SPL == 3 ? PW3[pi] : PW2[pi], bt = SPL == 3 ? PB3[pi] : PB2[pi];
                    const i32x4_t aw = (c >= NCX && ngate) ? Ah[wt] : Ai[wt];
                    if (c < NCX) accx[c < NCX ? c : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aw), __builtin_bit_cast(bf16x8_t, Bf[bt]), accx[c < NCX ? c : 0], 0, 0, 0);
                    else acch[c >= NCX ? c - NCX : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aw), __builtin_bit_cast(bf16x8_t, Bf[bt]), acch[c >= NCX ? c - NCX : 0], 0, 0, 0);
                }
            }
        }
    };
    tile_load(0);
    dg_load(0, gi0, gh0);
    for (int tile = 0; tile < ntiles; tile += 2) {
        // (two tiles per trip: the register sets of the gate-derivative column alternate with compile-time indices)
#pragma unroll
        for (int par = 0; par < 2; par++) {
            const int tl = tile + par;
            if (tl >= ntiles) break;
            const bool hz = hzero(tl);
            tile_store(hz);                                         // (waits for the tile's loads)
            lds_barrier_dw();                                       // the tile's terms are in LDS
            if (tl + 1 < ntiles) {                                  // travel underneath this tile's MFMAs
                tile_load(tl + 1);
                if (par == 0) dg_load(tl + 1, gi1, gh1); else dg_load(tl + 1, gi0, gh0);
            }
            if (par == 0) tile_mfma(hz, gi0, gh0); else tile_mfma(hz, gi1, gh1);
            lds_barrier_dw();                                       // every wave has read its fragments: the buffer may be rewritten
        }
    }
    if (jok && !(a.dbg & 1)) {
        const osk::rsrc_t rwx = osk::make_rsrc(a.dWih, (uint32_t)a.H3 * (uint32_t)a.Kx * 4u);
        const osk::rsrc_t rwh = osk::make_rsrc(a.dWhh, (uint32_t)a.H3 * (uint32_t)a.H * 4u);
#pragma unroll
        for (int c = 0; c < NCX; c++) {
            const int k = c * 32 + li;
            if (k < a.Kx) {
                const uint32_t vo = (uint32_t)((4 * kk) * a.Kx + k) * 4u;
#pragma unroll
                for (int e = 0; e < 16; e++) buf_atomic_add(accx[c][e], rwx, vo, (uint32_t)((j0 + (e & 3) + 8 * (e >> 2)) * a.Kx) * 4u);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint32_t vo = (uint32_t)((4 * kk) * a.H + c * 32 + li) * 4u;
#pragma unroll
            for (int e = 0; e < 16; e++) buf_atomic_add(acch[c][e], rwh, vo, (uint32_t)((j0 + (e & 3) + 8 * (e >> 2)) * a.H) * 4u);
        }
        bsi += __shfl_xor(bsi, 32, 64);              // the two row halves of the same gate unit
        bsh += __shfl_xor(bsh, 32, 64);
        if (kk == 0) {
            global_fadd(a.dbih, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsi);
            global_fadd(a.dbhh, (uint32_t)a.H3 * 4u, (uint32_t)(j0 + li), bsh);
        }
    }
}

// column sums of columns [n0, N) of a [R][N] row-major matrix into dst (bias gradients): 64 columns x 4 row-lanes per
// workgroup, 256 rows per workgroup (64 loads per thread, eight in flight), one atomic per column per workgroup
// copy_src / copy_n: block (0, 0) also copies copy_n floats copy_src -> dst (the r and z thirds of b_hh's gradient equal b_ih's)
__global__ void colsum_kernel(size_t R, int N, int n0, const float *src /* [R][ld], column n + shift */, int ld, int shift, float *dst, const float *copy_src, int copy_n)
{
    __shared__ float part[4][64];
    if (blockIdx.x == 0 && blockIdx.y == 0 && copy_src)
        for (int i = threadIdx.x; i < copy_n; i += blockDim.x) dst[i] = copy_src[i];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = n0 + blockIdx.x * 64 + cx;
    const size_t r0 = (size_t)blockIdx.y * 256, r1 = r0 + 256 < R ? r0 + 256 : R;
    float s = 0.f;
    if (n < N) {
#pragma unroll 8
        for (size_t r = r0 + ry; r < r1; r += 4) s += __builtin_nontemporal_load(src + r * ld + n + shift);
    }
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) atomicAdd(&dst[n], part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}

// [T][B][F] -> [B][T][F]
__global__ void permute_tb_kernel(int B, int T, int F, const float *src, float *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)B * T * F;
    if (i >= n) return;
    const int f = i % F;
    const size_t bt = i / F;
    const int t = bt % T, b = bt / T;
    dst[i] = src[((size_t)t * B + b) * F + f];
}

// fused Adam (torch.optim.Adam defaults, gru_train.py:219): one pass over the flat parameter / gradient vectors
// err: the context's error word (launch.hpp).  While a stacked launch's lost producer is unreported the gradients behind it are
// NaN-poisoned: the update is skipped (weights and moments unchanged), so an asynchronous caller (OS_GRU_STACK=2) never steps a model
// on them (gru/gru_train.py:247-249 would).
__global__ void adam_kernel(size_t n, float *w, const float *g, float *m, float *v, float lr, float b1, float b2, float eps,
                            float bc1, float bc2, const int32_t *err)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // (the device twin of the error word)
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    w[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace ost

using namespace ost;


// picks the accumulator count for the input width: the smallest NC in {2, 4, 6} that covers K in one pass, else 6 per pass
// (dw2_kernel); the fallback dw_kernel stops at 4 per pass -- with 6 accumulators its register-staged X tile spilled (540 B of
// scratch per lane), and it is what the LAST, partial batch of an epoch reaches (B % 32 != 0 with the batch_first input,
// gru/gru_train.py:199-201 keeps that batch): two passes over the rows for a 188-wide layer there, no scratch
static void launch_dw(const DwArgs &d, int K, dim3 grid, hipStream_t s)
{
    const int nkc = (K + 31) / 32;
    // dw2_kernel: 16-byte pieces (K % 4), X within one 32-bit buffer, and for the batch_first input whole 32-window tiles
    const bool v4 = (K & 3) == 0 && (size_t)d.T * d.B * K * 4 < ((size_t)1 << 31) &&
                    (!d.x_btf || (d.B % DW_TR == 0 && d.rows_per_slice % DW_TR == 0 && (d.r_begin - d.x_row_shift) % DW_TR == 0));
    if (v4) {
        if (nkc <= 2) hipLaunchKernelGGL((dw2_kernel<2>), grid, dim3(256), 0, s, d);
        else if (nkc <= 4) hipLaunchKernelGGL((dw2_kernel<4>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((dw2_kernel<6>), grid, dim3(256), 0, s, d);
    } else {
        if (nkc <= 2) hipLaunchKernelGGL((dw_kernel<false, 2>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((dw_kernel<false, 4>), grid, dim3(256), 0, s, d);
    }
}


struct os_train_state {
    float *act;   size_t act_floats;     // saved activations: L x 5 x [T][B][H]
    float *seq;   size_t seq_floats;     // SoA layer outputs [L][T][H][B] (forward inputs of the next layer)
    float *xs;    size_t xs_floats;      // SoA copy of the input [T][I][B]
    float *dg;    size_t dg_floats;      // gate derivatives [T][B][4H]: da_r | da_z | da_n | da_n * r
    float *dxy;   size_t dxy_floats;     // dx ping-pong [T][B][max(K,H)] x 2 + dh_T [B][H]
    float *dhx;   size_t dhx_floats;     // bwd_sweep_wide_kernel: dh exchange buffers, L x [T][B][H]
    bool sweep_wide_attr_set;
    float *wT;    size_t wT_floats;      // transposed-packed weights
    float *lossp; size_t lossp_floats;   // loss_kernel: per-workgroup partial sums + ticket counter
    int B, T;
    // weight-gradient reductions of layer l run on a side stream underneath the backward sweep of layer l-1 (they only
    // depend on layer l's gate derivatives): non-blocking stream + events, created on first use
    hipStream_t side;
    hipEvent_t ev_fork, ev_sweep[17], ev_dw[17];
    bool sweep_stack_attr_set;           // hipFuncSetAttribute done for bwd_sweep_stack_kernel on this context's device
    bool side_ready;
};

static int train_side_stream(os_ctx *ctx, os_train_state *ts)
{
    if (ts->side_ready) return 0;
    OS_HIP(ctx, hipStreamCreateWithFlags(&ts->side, hipStreamNonBlocking));
    OS_HIP(ctx, hipEventCreateWithFlags(&ts->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 17; i++) {
        OS_HIP(ctx, hipEventCreateWithFlags(&ts->ev_sweep[i], hipEventDisableTiming));
        OS_HIP(ctx, hipEventCreateWithFlags(&ts->ev_dw[i], hipEventDisableTiming));
    }
    ts->side_ready = true;
    return 0;
}

static os_train_state *train_state(os_ctx *ctx)
{
    if (!ctx->train) {
        os_train_state *t = (os_train_state *)calloc(1, sizeof(os_train_state));
        if (!t) return nullptr;
        ctx->train = t;
    }
    return (os_train_state *)ctx->train;
}

void os_train_destroy(os_ctx *ctx)
{
    os_train_state *t = (os_train_state *)ctx->train;
    if (!t) return;
    float *bufs[] = {t->act, t->seq, t->xs, t->dg, t->dxy, t->dhx, t->wT, t->lossp};
    for (float *b : bufs)
        if (b) (void)hipFree(b);
    if (t->side_ready) {
        (void)hipStreamSynchronize(t->side);
        (void)hipStreamDestroy(t->side);
        (void)hipEventDestroy(t->ev_fork);
        for (int i = 0; i < 17; i++) { (void)hipEventDestroy(t->ev_sweep[i]); (void)hipEventDestroy(t->ev_dw[i]); }
    }
    free(t);
    ctx->train = nullptr;
}

extern "C" {

size_t os_gru_train_ws_floats(const os_gru_dims *d, int32_t B, int32_t T)
{
    if (!d || B <= 0 || T <= 0) return 0;
    return (size_t)d->num_layers * 5 * (size_t)T * B * d->hidden_size;      // r, z, n, gh_n, h_t per layer, [T][B][H] each
}

// ws == nullptr: activations go to context scratch (os_gru_forward_train); otherwise to the caller's workspace
static int forward_train_impl(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, float *ws, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_gru_forward_train: call os_gru_load first");
    if (B <= 0 || T <= 0 || !x || !out) return os_fail(ctx, -2, "os_gru_forward_train: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_train_state *ts = train_state(ctx);
    if (!ts) return os_fail(ctx, -13, "os_gru_forward_train: cannot create training state");
    hipStream_t s = (hipStream_t)stream;
    const os_gru_dims &d = ctx->gru;
    const int H = d.hidden_size, L = d.num_layers, I = d.input_size;
    const size_t tbh = (size_t)T * B * H;
    float *act = ws;
    if (!act) {
        if (os_ensure_scratch(ctx, &ts->act, &ts->act_floats, (size_t)L * 5 * tbh)) return -10;
        act = ts->act;
        ts->B = B; ts->T = T;
    }
    if (os_ensure_scratch(ctx, &ts->seq, &ts->seq_floats, (size_t)L * tbh)) return -10;
    // small batches (the reference trains at 64, gru/gru_train.py:36): the whole forward as ONE layer-pipelined launch (gru_stack_kernel)
    const bool stack = os_gru_stack_eligible(ctx, B, T, I, H, L);
    const bool wide = stack && L >= 2 && os_gru_wide_eligible(ctx, B, T, I, H, L);      // gru_wide_kernel.hip
    // layer 0 reads the caller's (B, T, I) tensor itself where its kernel can (no SoA copy of the input)
    const bool x_direct = wide || (!stack && os_gru_layer_takes_btf(ctx, B, T, I, H));
    if (!x_direct) {
        if (os_ensure_scratch(ctx, &ts->xs, &ts->xs_floats, (size_t)T * I * B)) return -10;
        int rc = os_pack_stream(ctx, B, T, I, x, ts->xs, stream);
        if (rc) return rc;
    }
    size_t woff = 0;
    const float *in = x_direct ? x : ts->xs;
    osg::LayerArgs la[16];
    for (int l = 0; l < L; l++) {
        const int K = l == 0 ? I : H;
        osg::LayerArgs &a = la[l];
        a.B = B; a.T = T; a.K = K; a.H = H; a.KPx = (K + 1) / 2; a.KPh = H / 2;
        a.xs = in; a.xs_btf = (l == 0 && x_direct) ? 1 : 0; a.w = ctx->gru_packed + woff;
        a.seq_out = ts->seq + (size_t)l * tbh;
        a.h_last = nullptr;
        float *base = act + (size_t)l * 5 * tbh;
        a.sv_r = base; a.sv_z = base + tbh; a.sv_n = base + 2 * tbh; a.sv_g = base + 3 * tbh; a.sv_h = base + 4 * tbh;
        {
            // development (timing ablation only, OS_TRAIN_DBG_NOSAVE=1): the forward without its saved-activation stores -- an upper bound
            // on what ANY re-layout of those stores could gain (the backward then reads stale activations: wrong gradients)
            static const bool nosave = getenv("OS_TRAIN_DBG_NOSAVE") != nullptr;
            if (nosave) a.sv_r = a.sv_z = a.sv_n = a.sv_g = a.sv_h = nullptr;
        }
        if (!stack && os_gru_launch_layer(ctx, a, s)) return -10;
        in = a.seq_out;
        woff += os_layer_packed_floats(K, H);
    }
    if (stack && wide) {
        // four CUs per (layer, tile): the saved h stream [T][B][H] is the exchange buffer; the head's [H][B] comes out as h_last
        osg::WideArgs wa;
        wa.n = L; wa.tiles = (B + 31) / 32; wa.B = B; wa.T = T; wa.K0 = I; wa.xs0 = x; wa.xs0_btf = 1;
        for (int l = 0; l < L; l++) {
            wa.w[l] = la[l].w;
            wa.hseq[l] = la[l].sv_h ? la[l].sv_h : ts->seq + (size_t)l * tbh;      // (OS_TRAIN_DBG_NOSAVE: the SoA buffer's memory, row-major)
            wa.sv_r[l] = la[l].sv_r; wa.sv_z[l] = la[l].sv_z; wa.sv_n[l] = la[l].sv_n; wa.sv_g[l] = la[l].sv_g;
            wa.h_last[l] = l == L - 1 ? ts->seq + (size_t)(L - 1) * tbh + (size_t)(T - 1) * H * B : nullptr;
        }
        const int rc = os_gru_launch_wide(ctx, wa, la[0].sv_r != nullptr, s);
        if (rc) return rc;
    } else if (stack) {
        const int rc = os_gru_launch_stack(ctx, la, L, s);
        if (rc) return rc;
    }
    // head on h_T of the top layer: the SoA sequence's last step is [H][B]
    const float *top = ts->seq + (size_t)(L - 1) * tbh + (size_t)(T - 1) * H * B;
    const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
    return os_gru_head_launch(ctx, B, top, fcw, out, s);
}

int os_gru_forward_train(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (const int rcp = os_stack_pending(ctx, "os_gru_forward_train")) return rcp;
    return forward_train_impl(ctx, B, T, x, out, nullptr, stream);
}

int os_gru_forward_train_ws(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, float *ws, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!ws) return os_fail(ctx, -2, "os_gru_forward_train_ws: null workspace");
    if (const int rcp = os_stack_pending(ctx, "os_gru_forward_train_ws")) return rcp;
    return forward_train_impl(ctx, B, T, x, out, ws, stream);
}

int os_gru_loss(os_ctx *ctx, int32_t B, const float *out, const float *y, float *target, float *dout, float *loss,
                void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!ctx->gru_loaded || B <= 0 || !out || !y || !dout || !loss) return os_fail(ctx, -2, "os_gru_loss: bad argument");
    const int C = ctx->gru.num_classes;
    if (C % 2) return os_fail(ctx, -4, "os_gru_loss: num_classes must be even (state | error bands)");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    os_train_state *ts = train_state(ctx);
    if (!ts) return os_fail(ctx, -13, "os_gru_loss: cannot create training state");
    if (!ts->lossp) {
        if (os_ensure_scratch(ctx, &ts->lossp, &ts->lossp_floats, LOSS_MAXBLK + 1)) return -10;
        OS_HIP(ctx, hipMemsetAsync(ts->lossp, 0, (LOSS_MAXBLK + 1) * sizeof(float), s));       // the ticket counter, once
    }
    const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_MISC, s, "loss_kernel");
    {
        const int nblk = (B * C + 255) / 256;
        hipLaunchKernelGGL(loss_kernel, dim3(nblk < LOSS_MAXBLK ? nblk : LOSS_MAXBLK), dim3(256), 0, s, B, C, out, y, target, dout, loss, ts->lossp);
    }
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

static int backward_impl(os_ctx *ctx, const os_gru_dims &d, const float *w_flat, const float *act, int32_t B, int32_t T,
                         const float *x, const float *out, const float *dout, float *grad_flat, float *dx_out, void *stream)
{
    os_train_state *ts = train_state(ctx);
    if (!ts) return os_fail(ctx, -13, "os_gru_backward: cannot create training state");
    if (!x || !out || !dout || !grad_flat || !act || !w_flat) return os_fail(ctx, -2, "os_gru_backward: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int H = d.hidden_size, L = d.num_layers, I = d.input_size, C = d.num_classes, H3 = 3 * H;
    const size_t tbh = (size_t)T * B * H, nparam = os_gru_param_count(&d);
    const int Kmax = I > H ? I : H;
    // bwd_sweep_kernel addresses the saved activations, the gate derivatives and dx through buffer descriptors whose sizes
    // and per-step offsets are 32-bit byte counts: past 4 GiB they would wrap, out-of-range loads would read 0 and stores
    // be dropped -- silently wrong gradients.  Refuse instead (split the batch: the gradients of the parts add up).
    if ((size_t)T * B * 4 * H * sizeof(float) >= ((size_t)1 << 32) || (size_t)T * B * Kmax * sizeof(float) >= ((size_t)1 << 32))
        return os_fail(ctx, -2, "os_gru_backward: T*B*4*hidden_size (or T*B*input_size) floats reach 4 GiB, beyond the backward "
                                "sweep's 32-bit buffer offsets; split the batch");
    // small batches at H = 128: every layer's sweep in ONE launch, layers pipelined through progress counters (bwd_sweep_stack_kernel);
    // each layer then keeps its own gate-derivative and dx buffers
    const bool stacked = ctx->tune_gru_stack != 0 && L >= 2 && L <= 8 && H == 128 && I <= 192 && ((B + 31) / 32) * L <= ctx->cu_count &&
                         ctx->tune_sweep_wr == 32 && (ctx->tune_sweep_nw == 0 || ctx->tune_sweep_nw == 8) && !dx_out;
    // weight gradients on a side stream underneath the next layer's sweep: where CUs are idle (at most half a chip of 32-row tiles), a
    // dW launch is long enough to be worth two event hand-overs (measured: 512 windows 1.01 -> 0.96 ms per step, 64 windows 0.90 -> 0.92)
    // and the sweeps are separate launches (behind the stacked sweep there is nothing left to hide under: 0.81 against 0.74 ms)
    const bool overlap = L > 1 && (ctx->tune_train_overlap > 0 ||
                                   (ctx->tune_train_overlap < 0 && !stacked && 2 * ((B + 31) / 32) <= ctx->cu_count && (size_t)T * B >= 2048));
    if (os_ensure_scratch(ctx, &ts->dg, &ts->dg_floats, (stacked ? L : overlap ? 2 : 1) * (size_t)T * B * 4 * H)) return -10;
    if (overlap && train_side_stream(ctx, ts)) return -10;
    // ... on four CUs per (layer, tile) where gru_wide_kernel's shape conditions hold (bwd_sweep_wide_kernel)
    const bool wide_bwd = stacked && os_gru_wide_eligible(ctx, B, T, I, H, L);
    if (wide_bwd && os_ensure_scratch(ctx, &ts->dhx, &ts->dhx_floats, (size_t)L * tbh)) return -10;
    const int ndx = stacked ? L : 2;
    if (os_ensure_scratch(ctx, &ts->dxy, &ts->dxy_floats, ndx * (size_t)T * B * Kmax + (size_t)B * H)) return -10;
    // transposed-packed weights (re-done every call: parameters change every optimiser step)
    size_t wT_total = 0;
    for (int l = 0; l < L; l++) {
        const int K = l == 0 ? I : H;
        wT_total += (size_t)((K + 31) / 32 + H / 32) * (H3 / 2) * 64;
    }
    if (os_ensure_scratch(ctx, &ts->wT, &ts->wT_floats, wT_total)) return -10;
    const size_t rows = (size_t)T * B;
    size_t poff[17], wToff[17];
    {
        size_t po = 0, wo = 0;
        for (int l = 0; l < L; l++) {
            const int K = l == 0 ? I : H;
            poff[l] = po; wToff[l] = wo;
            po += (size_t)H3 * K + (size_t)H3 * H + 2 * (size_t)H3;
            wo += (size_t)((K + 31) / 32 + H / 32) * (H3 / 2) * 64;
        }
    }
    {
        PackTAll pa;
        pa.n = 0;
        int maxch = 1;
        for (int l = 0; l < L && pa.n + 2 <= 32; l++) {
            const int K = l == 0 ? I : H;
            const float *Wih = w_flat + poff[l], *Whh = Wih + (size_t)H3 * K;
            float *wihT = ts->wT + wToff[l], *whhT = wihT + (size_t)((K + 31) / 32) * (H3 / 2) * 64;
            pa.K[pa.n] = K; pa.chunks[pa.n] = (K + 31) / 32; pa.W[pa.n] = Wih; pa.dst[pa.n] = wihT; pa.n++;
            pa.K[pa.n] = H; pa.chunks[pa.n] = H / 32; pa.W[pa.n] = Whh; pa.dst[pa.n] = whhT; pa.n++;
            if ((K + 31) / 32 > maxch) maxch = (K + 31) / 32;
            if (H / 32 > maxch) maxch = H / 32;      // (round 5, found by tools/fuzz_shapes.py: a one-layer model with fewer input chunks
                                                     // than hidden chunks left W_hh^T's upper chunks unpacked: wrong gradients)
        }
        const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_MISC, s, "pack_T_all_kernel");
        hipLaunchKernelGGL(pack_T_all_kernel, dim3(maxch, 16, pa.n), dim3(256), 0, s, pa, H3, grad_flat, nparam);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
    }
    // ---- head ----
    const size_t fc_off = nparam - ((size_t)C * H + C);
    const float *fcw = w_flat + fc_off;
    float *dhT = ts->dxy + ndx * (size_t)T * B * Kmax;
    const float *hT = act + ((size_t)(L - 1) * 5 + 4) * tbh + (size_t)(T - 1) * B * H;   // sv_h of the top layer, t = T-1
    {
        const size_t lds = (size_t)(64 * ((C + 31) / 32 * 32) + 64 * (H + 1)) * sizeof(float);
        const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_MISC, s, "head_backward_kernel");
        hipLaunchKernelGGL(head_backward_kernel, dim3((B + 63) / 64), dim3(512), lds, s, B, H, C, d.use_sigmoid, out, dout, hT,
                           fcw, dhT, grad_flat + fc_off, grad_flat + fc_off + (size_t)C * H);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
    }
    // ---- layers, top to bottom ----
    auto dx_of = [&](int l) { return ts->dxy + (size_t)(stacked ? l : (l & 1)) * T * B * Kmax; };
    auto dg_of = [&](int l) { return ts->dg + (stacked ? (size_t)l : overlap ? (size_t)(l & 1) : 0) * (size_t)T * B * 4 * H; };
    auto sweep_args = [&](int l) {
        const int K = l == 0 ? I : H;
        float *wihT = ts->wT + wToff[l], *whhT = wihT + (size_t)((K + 31) / 32) * (H3 / 2) * 64;
        const float *base = act + (size_t)l * 5 * tbh;
        SweepArgs a;
        a.B = B; a.T = T; a.K = K; a.H = H;
        a.need_dx = (l > 0 || dx_out) ? 1 : 0;
        a.sv_r = base; a.sv_z = base + tbh; a.sv_n = base + 2 * tbh; a.sv_g = base + 3 * tbh; a.sv_h = base + 4 * tbh;
        a.dy = l == L - 1 ? nullptr : dx_of(l + 1); a.dy_last = (l == L - 1) ? dhT : nullptr;
        a.wihT = wihT; a.whhT = whhT; a.dg4 = dg_of(l);
        a.dx = dx_of(l);
        a.flag_prev = nullptr; a.flag_mine = nullptr;
        return a;
    };
    hipStream_t sw = overlap ? ts->side : s;             // stream of the weight-gradient kernels
    if (overlap) {
        OS_HIP(ctx, hipEventRecord(ts->ev_fork, s));     // the side stream starts behind the gradient clear / head backward
        OS_HIP(ctx, hipStreamWaitEvent(ts->side, ts->ev_fork, 0));
    }
    for (int l = L - 1; l >= 0; l--) {
        const int K = l == 0 ? I : H;
        // gate derivatives: two buffer sets by layer parity when overlapping, so that the sweep of layer l-1 does not
        // overwrite what the reductions of layer l are still reading; layer l reuses the set of layer l+2
        float *dg4 = dg_of(l);
        if (overlap && !stacked && l + 2 < L) OS_HIP(ctx, hipStreamWaitEvent(s, ts->ev_dw[l + 2], 0));
        const float *base = act + (size_t)l * 5 * tbh;
        const SweepArgs a = sweep_args(l);
        const int RB = H <= 64 ? 2 : 1;
        const int BM = 32 * RB;
        const size_t lds = (size_t)(BM * (H + 1) + BM * (4 * H + 1)) * sizeof(float);
        if (!ctx->sweep_attr_set) {            // per context (= per device), not process-global
            OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_kernel<1, 8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->sweep_attr_set = true;
        }
        dim3 grid((B + BM - 1) / BM);
        int nw = (RB == 1 && ((a.need_dx ? (K + 31) / 32 : 0) + H / 32 >= 8 || (!a.need_dx && H / 32 >= 4))) ? 8 : 4;   // enough work items for eight waves
        if (ctx->tune_sweep_nw) nw = ctx->tune_sweep_nw == 8 && RB == 1 ? 8 : 4;
        if (stacked && wide_bwd) {
            if (l == L - 1) {                                  // four CUs per (layer, tile): bwd_sweep_wide_kernel
                SweepWideArgs wa;
                wa.n = L; wa.tiles = (B + 31) / 32;
                const size_t nfl = (size_t)wa.n * wa.tiles * 4;
                if (ctx->stack_flags_n < nfl) {
                    if (ctx->stack_flags) OS_HIP(ctx, hipFree(ctx->stack_flags));
                    ctx->stack_flags = nullptr; ctx->stack_flags_n = 0;
                    OS_HIP(ctx, hipMalloc((void **)&ctx->stack_flags, nfl * sizeof(uint32_t)));
                    ctx->stack_flags_n = nfl;
                }
                wa.flags = ctx->stack_flags;
                wa.err = ctx->stack_err_dev; wa.err_local = ctx->stack_err_local; wa.max_polls = ctx->stack_max_polls;
                wa.drop_y = ctx->stack_dbg_drop_layer >= 0 ? L - 1 - ctx->stack_dbg_drop_layer : -1; wa.drop_step = ctx->stack_dbg_drop_step;
                OS_HIP(ctx, hipMemsetAsync(wa.flags, 0, nfl * sizeof(uint32_t), s));
                for (int y = 0; y < L; y++) { wa.layer[y] = sweep_args(L - 1 - y); wa.dhx[y] = ts->dhx + (size_t)y * tbh; }
                if (!ts->sweep_wide_attr_set) {
                    OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    ts->sweep_wide_attr_set = true;
                }
                const size_t lds_w = (size_t)(32 * (H + 1) + 32 * (4 * H + 1) + 8 * 4 * 64 * 4) * sizeof(float);
                const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_SWEEP, s, "bwd_sweep_wide_kernel");
                hipLaunchKernelGGL(bwd_sweep_wide_kernel, dim3(8u * (unsigned)((wa.tiles + 7) / 8 * 4 * L)), dim3(512), lds_w, s, wa);
                os_prof_end(ctx, slot, s);
                if (const int rcv = os_stack_verify(ctx, s, "bwd_sweep_wide_kernel")) return rcv;
            }
        } else if (stacked) {
            if (l == L - 1) {                                  // every layer's sweep in this one launch; the iterations below only reduce
                SweepStackArgs sa;
                sa.n = L; sa.tiles = (B + 31) / 32;
                const size_t nfl = (size_t)sa.n * sa.tiles;
                if (ctx->stack_flags_n < nfl) {
                    if (ctx->stack_flags) OS_HIP(ctx, hipFree(ctx->stack_flags));
                    ctx->stack_flags = nullptr; ctx->stack_flags_n = 0;
                    OS_HIP(ctx, hipMalloc((void **)&ctx->stack_flags, nfl * sizeof(uint32_t)));
                    ctx->stack_flags_n = nfl;
                }
                sa.flags = ctx->stack_flags;
                sa.err = ctx->stack_err_dev; sa.err_local = ctx->stack_err_local; sa.max_polls = ctx->stack_max_polls;
                // (the debug knob names a LAYER of the model; this launch's row 0 is the top layer)
                sa.drop_y = ctx->stack_dbg_drop_layer >= 0 ? L - 1 - ctx->stack_dbg_drop_layer : -1; sa.drop_step = ctx->stack_dbg_drop_step;
                OS_HIP(ctx, hipMemsetAsync(sa.flags, 0, nfl * sizeof(uint32_t), s));
                for (int y = 0; y < L; y++) sa.layer[y] = sweep_args(L - 1 - y);
                if (!ts->sweep_stack_attr_set) {
                    OS_HIP(ctx, hipFuncSetAttribute((const void *)bwd_sweep_stack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    ts->sweep_stack_attr_set = true;
                }
                const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_SWEEP, s, "bwd_sweep_stack_kernel");
                sa.y0 = 0;
                static const bool one_by_one = getenv("OS_SWEEP_STACK_DBG") != nullptr;   // debugging: the same kernel, one layer per launch
                if (one_by_one) {
                    for (int y = 0; y < L; y++) { sa.y0 = y; hipLaunchKernelGGL(bwd_sweep_stack_kernel, dim3(sa.tiles, 1), dim3(512), lds, s, sa); }
                } else
                hipLaunchKernelGGL(bwd_sweep_stack_kernel, dim3(sa.tiles, L), dim3(512), lds, s, sa);
                os_prof_end(ctx, slot, s);
                if (const int rcv = os_stack_verify(ctx, s, "bwd_sweep_stack_kernel")) return rcv;
            }
        } else {
            // resident leading k-pairs: one work item per wave
            const int nitems = ((a.need_dx ? (K + 31) / 32 : 0) + H / 32) * ((!a.need_dx && 8 > H / 32) ? 2 : 1);
            const int wr = (nw == 8 && H == 128 && nitems <= 8 && ctx->tune_sweep_wr != 0) ? 32 : 0;      // OS_SWEEP_WR=0: all of it streamed
            const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_SWEEP, s,
                                           RB == 2 ? "bwd_sweep_kernel<2,4>" : wr == 32 ? "bwd_sweep_kernel<1,8,32>" :
                                           nw == 8 ? "bwd_sweep_kernel<1,8>" : "bwd_sweep_kernel<1,4>");
            if (RB == 2) hipLaunchKernelGGL((bwd_sweep_kernel<2, 4>), grid, dim3(256), lds, s, a);
            else if (wr == 32) hipLaunchKernelGGL((bwd_sweep_kernel<1, 8, 32>), grid, dim3(512), lds, s, a);
            else if (nw == 8) hipLaunchKernelGGL((bwd_sweep_kernel<1, 8>), grid, dim3(512), lds, s, a);
            else hipLaunchKernelGGL((bwd_sweep_kernel<1, 4>), grid, dim3(256), lds, s, a);
            os_prof_end(ctx, slot, s);
        }
        OS_HIP(ctx, hipGetLastError());
        if (overlap) {
            OS_HIP(ctx, hipEventRecord(ts->ev_sweep[l], s));
            OS_HIP(ctx, hipStreamWaitEvent(sw, ts->ev_sweep[l], 0));
        }
        // ---- weight and bias gradients: one launch each for W_ih (+b_ih) and W_hh (+b_hh) ----
        float *gWih = grad_flat + poff[l], *gWhh = gWih + (size_t)H3 * K, *gbih = gWhh + (size_t)H3 * H, *gbhh = gbih + H3;
        // both products in one launch (dw3_kernel) when the tiles are whole and the accumulators fit two workgroups per CU
        // An input of 129..192 columns (the 188-wide first layer) takes the ten-accumulator form on 16-row tiles.  (Tried and not
        // kept: its leading 128 columns in dw3_kernel<4,4> and the rest through dw2_kernel<2>: 0.177 + 0.071 ms against 0.237 for
        // the two separate products -- two MFMAs per gate-derivative load leave the second launch waiting on memory.)
        const bool wide = K > 128 && K <= 192 && H == 128;
        const int ncx = wide ? 6 : (K + 31) / 32 <= 2 ? 2 : 4, nchh = H / 32;
        // rows per slice: OS_DW_RPS, or about 32 slices, between one 32-row tile and 512 rows.  512 is the training batch's tuned value
        // (2 x 240 workgroups: two per CU); small batches want short slices (the reference's batch of 64: 640 rows = 20 tiles, one per
        // workgroup instead of sixteen in a row: 90 -> 25 us per launch) but not one tile each at 512 windows: every slice adds a whole
        // dW with atomics
        int rps = ctx->tune_dw_rps;
        if (rps <= 0) {
            rps = (int)(((rows + 31) / 32 + DW_TR - 1) / DW_TR) * DW_TR;
            rps = rps < DW_TR ? DW_TR : rps > 512 ? 512 : rps;
        }
        const bool fuse_dw = ctx->tune_dw_fused != 0 && (K <= 128 || (wide && ctx->tune_dw_fused != 2)) && (K & 3) == 0 && (nchh == 4 || (nchh == 2 && ncx == 2)) && B % DW_TR == 0 &&
                             rps % DW_TR == 0 && T > 1 && (size_t)T * B * (K > H ? K : H) * 4 < ((size_t)1 << 31);
        if (fuse_dw) {
            Dw3Args d;
            d.H3 = H3; d.Kx = K; d.H = H; d.rows = rows; d.rows_per_slice = rps; d.dG = dg4; d.ldg = 4 * H;
            d.B = B; d.T = T; d.Hp = base + 4 * tbh;
            if (l == 0) { d.X = x; d.x_btf = 1; }
            else { d.X = act + ((size_t)(l - 1) * 5 + 4) * tbh; d.x_btf = 0; }
            d.dWih = gWih; d.dWhh = gWhh; d.dbih = gbih; d.dbhh = gbhh;
            d.ngroups = (H3 / 32 + 3) / 4;
            d.dbg = ctx->tune_dw_dbg;
            const unsigned nslices = (unsigned)((rows + rps - 1) / rps);
            const dim3 grid(8u * ((nslices + 7) / 8) * (unsigned)d.ngroups);       // see the XCD-aware order in the kernel
            const int spl = (ctx->gru_split_bf16 & OS_GRU_SPLIT_TRAIN) ? (ctx->gru_split_bf16 & 3) : 0;      // opt-in: bf16 terms per fp32 operand
            const int dslot = os_prof_begin(ctx, OS_PHASE_TRAIN_DW, sw, spl == 3 ? "dw3_bf16_kernel<3>" : spl == 2 ? "dw3_bf16_kernel<2>" : "dw3_kernel");
            if (spl && rps % 32 == 0 && B % 32 == 0) {
#define OST_DWBF(NCX_, NCH_) { if (spl == 3) hipLaunchKernelGGL((dw3_bf16_kernel<NCX_, NCH_, 3>), grid, dim3(256), 0, sw, d); else hipLaunchKernelGGL((dw3_bf16_kernel<NCX_, NCH_, 2>), grid, dim3(256), 0, sw, d); }
                if (wide) OST_DWBF(6, 4) else if (nchh == 4 && ncx == 4) OST_DWBF(4, 4) else if (nchh == 4) OST_DWBF(2, 4) else OST_DWBF(2, 2)
#undef OST_DWBF
            }
            else if (wide) hipLaunchKernelGGL((dw3_kernel<6, 4, 16>), grid, dim3(256), 0, sw, d);
            else if (nchh == 4 && ncx == 4) hipLaunchKernelGGL((dw3_kernel<4, 4>), grid, dim3(256), 0, sw, d);
            else if (nchh == 4) hipLaunchKernelGGL((dw3_kernel<2, 4>), grid, dim3(256), 0, sw, d);
            else hipLaunchKernelGGL((dw3_kernel<2, 2>), grid, dim3(256), 0, sw, d);
            os_prof_end(ctx, dslot, sw);
            OS_HIP(ctx, hipGetLastError());
            if (overlap) OS_HIP(ctx, hipEventRecord(ts->ev_dw[l], sw));
        } else {
            // rows per slice (above): T*B / rps slices x 3H/32 gate chunks of waves
            DwArgs d1;
            d1.H3 = H3; d1.K = K; d1.r_begin = 0; d1.r_end = rows; d1.x_row_shift = 0; d1.x_valid_from = 0; d1.rows_per_slice = rps;
            d1.dG = dg4; d1.ldg = 4 * H; d1.nshift = 0; d1.dW = gWih; d1.db = gbih; d1.B = B; d1.T = T;
            if (l == 0) { d1.X = x; d1.x_btf = 1; }
            else { d1.X = act + ((size_t)(l - 1) * 5 + 4) * tbh; d1.x_btf = 0; }
            const int dslot = os_prof_begin(ctx, OS_PHASE_TRAIN_DW, sw, "dw_kernel");
            launch_dw(d1, K, dim3((H3 / 32 + 3) / 4, (unsigned)((rows + rps - 1) / rps)), sw);
            // recurrent weights: rows t >= 1 pair with h_{t-1}; the bias sum still runs over every row
            DwArgs d2 = d1;
            d2.K = H; d2.r_begin = (size_t)B; d2.x_row_shift = (size_t)B; d2.nshift = H; d2.X = base + 4 * tbh; d2.x_btf = 0;
            d2.dW = gWhh; d2.db = nullptr;
            // dw2_kernel (H % 4 == 0) with whole 32-row tiles in the first time step: the launch covers ALL rows, the t = 0 tiles
            // feed only the bias sum, and b_hh's gradient needs no separate column-sum launch
            const bool bias_in_dw = (H & 3) == 0 && (B % DW_TR) == 0 && (rps % DW_TR) == 0;
            if (bias_in_dw) {
                d2.r_begin = 0; d2.x_valid_from = (size_t)B; d2.db = gbhh;
                launch_dw(d2, H, dim3((H3 / 32 + 3) / 4, (unsigned)((rows + rps - 1) / rps)), sw);
            } else if (T > 1) launch_dw(d2, H, dim3((H3 / 32 + 3) / 4, (unsigned)((rows - B + rps - 1) / rps)), sw);
            os_prof_end(ctx, dslot, sw);
            // b_hh: dgh differs from dgi only in the n gate (da_n * r instead of da_n), so the r and z thirds of the two bias
            // gradients are the same sums: copy them from b_ih (complete after the dW_ih launch) and reduce the n third only
            if (!bias_in_dw) {
                dim3 cg((H + 63) / 64, (unsigned)((rows + 255) / 256));
                const int cslot = os_prof_begin(ctx, OS_PHASE_TRAIN_MISC, sw, "colsum_kernel");
                hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, sw, rows, H3, 2 * H, (const float *)dg4, 4 * H, H, gbhh, (const float *)gbih, 2 * H);
                os_prof_end(ctx, cslot, sw);
            }
            OS_HIP(ctx, hipGetLastError());
            if (overlap) OS_HIP(ctx, hipEventRecord(ts->ev_dw[l], sw));
        }
        // os_gru_backward_mark: layers l .. L-1 (and the head, done first) have their gradients in the flat vector from here on
        if (ctx->bwd_mark_event && l == ctx->bwd_mark_layer) OS_HIP(ctx, hipEventRecord((hipEvent_t)ctx->bwd_mark_event, sw));
    }
    if (overlap) OS_HIP(ctx, hipStreamWaitEvent(s, ts->ev_dw[0], 0));      // join: the side stream is in order
    if (dx_out) {
        // dx of layer 0 is [T][B][I]; the caller's layout is (B, T, I)
        const size_t n = (size_t)B * T * I;
        hipLaunchKernelGGL(permute_tb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, T, I, dx_of(0), dx_out);
        OS_HIP(ctx, hipGetLastError());
    }
    return 0;
}

int os_gru_backward(os_ctx *ctx, int32_t B, int32_t T, const float *x, const float *out, const float *dout,
                    float *grad_flat, float *dx_out, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_gru_backward: call os_gru_load first");
    os_train_state *ts = (os_train_state *)ctx->train;
    if (!ts || !ts->act || ts->B != B || ts->T != T) return os_fail(ctx, -5, "os_gru_backward: call os_gru_forward_train first (same B, T)");
    if (const int rcp = os_stack_pending(ctx, "os_gru_backward")) return rcp;
    return backward_impl(ctx, ctx->gru, ctx->gru_flat, ts->act, B, T, x, out, dout, grad_flat, dx_out, stream);
}

int os_gru_backward_mark(os_ctx *ctx, int32_t layer, void *event)
{
    OS_CHECK_CTX(ctx);
    if (layer < 0) return os_fail(ctx, -2, "os_gru_backward_mark: bad layer");
    ctx->bwd_mark_layer = layer;
    ctx->bwd_mark_event = event;
    return 0;
}

int os_gru_backward_ws(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, int32_t B, int32_t T, const float *x,
                       const float *out, const float *dout, const float *ws, float *grad_flat, float *dx, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!d || B <= 0 || T <= 0) return os_fail(ctx, -2, "os_gru_backward_ws: bad argument");
    const int H = d->hidden_size;
    if (H != 32 && H != 64 && H != 128) return os_fail(ctx, -4, "os_gru_backward_ws: hidden_size must be 32, 64 or 128");
    if (const int rcp = os_stack_pending(ctx, "os_gru_backward_ws")) return rcp;
    return backward_impl(ctx, *d, w_flat, ws, B, T, x, out, dout, grad_flat, dx, stream);
}

int os_adam_step(os_ctx *ctx, size_t n, float *w, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                 float eps, int32_t step, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!n || !w || !g || !m || !v || step < 1) return os_fail(ctx, -2, "os_adam_step: bad argument");
    if (const int rcp = os_stack_pending(ctx, "os_adam_step")) return rcp;
    OS_HIP(ctx, hipSetDevice(ctx->device));
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const int slot = os_prof_begin(ctx, OS_PHASE_TRAIN_MISC, (hipStream_t)stream, "adam_kernel");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, w, g, m, v, lr,
                       beta1, beta2, eps, bc1, bc2, (const int32_t *)ctx->stack_err_local);
    os_prof_end(ctx, slot, (hipStream_t)stream);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
