// gru_wide_kernel.hip -- the GRU layer stack of a SMALL batch (the reference trains at 64 windows, gru/gru_train.py:32-36) spread over
// FOUR compute units per (layer, 32-row tile) instead of one.
//
// gru_stack_kernel gives a (layer, tile) to one workgroup: at B = 64, L = 4 that is 8 of 256 CUs, each grinding through the whole gate
// GEMM of its tile -- 32 x 316 x 384 x 2 flop per step = 12.7 us on one CU's fp32 matrix pipes, 19.5 us per pipeline stage measured.
// Here the 128 hidden units of a layer are split into four groups of 32 (one 32-column chunk of the packed weight image each); group q of
// (layer l, tile) is its own workgroup on its own CU:
//   * its 8 waves are 8 slices of the reduction; a wave's weight fragments -- <= 12 k-pairs of W_ih and 8 of W_hh, three gates -- are
//     loaded ONCE and stay in 60 registers for all T steps: no weight traffic inside the recurrence;
//   * per step the four groups exchange their 32-unit slices of h_t through a row-major [T][B][H] buffer (the training forward's saved
//     h stream itself) with write-through stores and one progress counter per group; a group starts step t + 1 when the four counters of
//     its layer say h_t is complete, and takes x_{t+1} = h^{l-1}_{t+1} when the four counters of the layer below say so;
//   * a step is: publish h_t -> wait for the lower layer's counters, x_{t+1} tile -> LDS, its A fragments -> registers -> wait for this
//     layer's counters, request the h_t tile -> the INPUT half's MFMAs (36 / 24 per wave) while that tile travels -> h_t tile -> LDS ->
//     the recurrent half's 24 MFMAs -> exchange of the partial sums -> cell update; layer 0's input depends on nothing in the launch and
//     is requested a whole step ahead (the caller's (B, T, I) tensor is read in place, 16-byte pieces);
//   * the partial sums cross LDS once, as 16-byte pieces in a buffer that ALIASES the two tiles (dead once every wave holds its A
//     fragments); every wave sums and updates the two accumulator elements (rows) it owns, so the cell update runs on all eight waves;
//   * block index -> (tile, layer, group) puts every workgroup of a tile on ONE XCD (blockIdx % 8): a placement choice only, the
//     exchange goes through cache-bypassing accesses and does not depend on where a workgroup lands (measured with -DWD_SPREAD, a tile's
//     groups on four different XCDs: 16.0 - 16.3 k cycles per step against 15.5 k).
// Every wait is bounded exactly as in gru_stack_kernel (error word, NaN poisoning, -20 from the call: launch.hpp), and the kernel needs
// all its workgroups resident at once: 4 x layers x tiles <= CUs with at most 32 / (4 x layers) tiles per XCD (B <= 512 at four layers).
// Same arithmetic as the other exact-fp32 layer kernels up to the order of the eight partial sums.
#include "launch.hpp"

#include "gru_common.hpp"
#include "kf_device.hpp"
#include "gru_device.hpp"

namespace osg {

using osk::buf_load;
using osk::make_rsrc;
using osk::rsrc_t;

constexpr int WD_H = 128, WD_HS = WD_H + 1, WD_NKX = 12, WD_NKH = 8, WD_XCH = 8 * 4 * 16 * 64;
#ifndef WD_AUX
#define WD_AUX 17      // sc0 | sc1: system scope (development: 16 = sc1, agent scope)
#endif


__device__ __forceinline__ float wd_load_coh(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, WD_AUX));      // sc0 sc1: coherent across workgroups / XCDs
}
__device__ __forceinline__ void wd_store_coh(rsrc_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, WD_AUX);      // write-through
}
// The counters of this layer's four groups (lanes 0..3) and of the layer below (lanes 4..7), polled with system-scope loads by every
// wave; false after max_polls rounds.  Plain coherent loads / stores, not acquire / release atomics: those come with an L2 write-back
// before every release and an L2 invalidate behind every polling acquire, and everything that crosses workgroups here is moved with
// cache-bypassing accesses anyway (the publisher waits for its data stores' acknowledgements before it writes its counter).
__device__ __forceinline__ bool wd_wait8(rsrc_t rf, uint32_t own_off, uint32_t need_own, uint32_t prev_off, uint32_t need_prev, uint32_t max_polls, const int32_t *err_local)
{
    const int lane = threadIdx.x & 63;
    const uint32_t need = lane < 4 ? need_own : need_prev;
    const uint32_t off = (lane < 4 ? own_off : prev_off) + (uint32_t)(lane & 3) * 4u;
    for (uint32_t spin = 0; spin < max_polls; spin++) {
        uint32_t v = need;
        if (lane < 8 && need) v = __builtin_amdgcn_raw_buffer_load_b32(rf, off, 0u, WD_AUX);
        if (__builtin_amdgcn_ballot_w64(v < need) == 0) return true;
        if ((spin & 1023u) == 1023u && stack_lost_already(err_local)) return false;      // somebody else gave up: so do we
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__device__ __forceinline__ void wd_lost(int32_t *err, int32_t *err_local)
{
    if ((threadIdx.x & 63) == 0) {
        if (err_local) __hip_atomic_fetch_or(err_local, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err) __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool SAVE>
__global__ __launch_bounds__(512, 1) void gru_wide_kernel(const WideArgs a)
{
    // xch [8 waves][4 gates][4 quads][64 lanes][4] (the partial sums) ALIASES the two tiles ht [32][129] | xt [32][XS]: the tiles are dead once
    // every wave holds its A fragments in registers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    // block -> (tile, layer, group): blockIdx % 8 = XCD = tile % 8; inside an XCD the slots run (tile / 8, layer, group)
#ifdef WD_SPREAD      // development (tools/wide_ts.sh -DWD_SPREAD): consecutive blocks = the four groups of a (tile, layer), i.e. four different XCDs
    const int q = blockIdx.x & 3, l = (int)(blockIdx.x >> 2) % a.n, tile = (int)(blockIdx.x >> 2) / a.n;
#else
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = 4 * a.n;
    const int tile = (slot / per) * 8 + xcd, l = (slot % per) >> 2, q = slot & 3;
#endif
    if (tile >= a.tiles) return;
    const int B = a.B, T = a.T, K = l == 0 ? a.K0 : WD_H, KPx = (K + 1) / 2, XS = 2 * KPx + 1;
    float *ht = smem, *xt = smem + 32 * WD_HS, *xch = smem;
    const int row0 = tile * 32;

    // ---- this wave's weight fragments: chunk q, x k-pairs [xb, xb + nkx), h k-pairs [8 wave, 8 wave + 8) ----
    const float *wx = a.w[l] + (size_t)q * chunk_floats(KPx, WD_H / 2);
    const float *wh = wx + (size_t)KPx * 3 * 64;
    const float *bias = wh + (size_t)(WD_H / 2) * 3 * 64;
    const int xb = KPx * wave / 8, nkx = KPx * (wave + 1) / 8 - xb, hb = WD_NKH * wave;
    float wxr[WD_NKX][3], whr[WD_NKH][3];
#pragma unroll
    for (int j = 0; j < WD_NKX; j++)
#pragma unroll
        for (int g = 0; g < 3; g++) wxr[j][g] = j < nkx ? wx[(size_t)((xb + j) * 3 + g) * 64 + lane] : 0.f;
#pragma unroll
    for (int j = 0; j < WD_NKH; j++)
#pragma unroll
        for (int g = 0; g < 3; g++) whr[j][g] = wh[(size_t)((hb + j) * 3 + g) * 64 + lane];
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];

    const rsrc_t rf = make_rsrc(a.flags, (uint32_t)a.n * (uint32_t)a.tiles * 16u);
    const uint32_t own_off = (uint32_t)((l * a.tiles + tile) * 16), prev_off = l > 0 ? (uint32_t)(((l - 1) * a.tiles + tile) * 16) : 0u;
    bool lost = stack_lost_already(a.err_local);                      // latched: after one expired wait (anywhere in the launch) this workgroup stops waiting

    for (int i = threadIdx.x; i < 32 * WD_HS; i += 512) ht[i] = 0.f;   // h0 = 0 (gru/gru_model.py:27)
    for (int i = threadIdx.x; i < 32 * XS; i += 512) xt[i] = 0.f;      // (the pad column of an odd input width stays zero)
    __syncthreads();

    // tile of a row-major [B][128] step (a sibling's / the lower layer's h): thread -> (row = tid / 16, columns tid % 16 + 16 e)
    const int srow = threadIdx.x >> 4, sk = threadIdx.x & 15;
    const uint32_t rm_off = (uint32_t)(((size_t)(row0 + srow) * WD_H + sk) * 4);
    const uint32_t step_bytes = (uint32_t)B * WD_H * 4u;
    float vx[WD_NKX];
    const int k0 = threadIdx.x >> 5, g = row0 + li < B ? row0 + li : B - 1;
    // Layer 0's x_t depends on nothing inside the launch: its loads are issued a whole step ahead (behind the A-fragment reads of step
    // t - 1) and wait in registers, so the input -- first touched here, straight from HBM when it is the caller's tensor -- is never on a
    // step's critical path.
    auto x0_issue = [&](int t) {
        if (a.xs0_btf) {
            // the caller's (B, T, K) tensor itself: thread -> (row tid / 16, inputs tid % 16 + 16 e), 64-byte runs of a row; rows past the
            // batch fall outside the descriptor (read zero), inputs past K are masked (they would be the next row's)
            const rsrc_t r = make_rsrc(a.xs0, (uint32_t)B * (uint32_t)T * (uint32_t)K * 4u);
            if ((K & 3) == 0) {
                // 16-byte pieces (every width the reference uses: 60, 188): three loads per thread, piece sk + 16 e of the row
                const uint32_t vo = (uint32_t)((((size_t)(row0 + srow) * T + t) * K + 4 * sk) * 4);
#pragma unroll
                for (int e = 0; e < WD_NKX / 4; e++) {
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (4 * (sk + 16 * e) < K) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (uint32_t)(e * 256), 0));
                    vx[4 * e] = v[0]; vx[4 * e + 1] = v[1]; vx[4 * e + 2] = v[2]; vx[4 * e + 3] = v[3];
                }
            } else {
                const uint32_t vo = (uint32_t)((((size_t)(row0 + srow) * T + t) * K + sk) * 4);
#pragma unroll
                for (int e = 0; e < WD_NKX; e++) vx[e] = sk + 16 * e < K ? buf_load(r, vo, (uint32_t)(e * 64)) : 0.f;
            }
        } else {
            // the caller's SoA stream [T][K][B]: thread -> (input k0 + 16 e, row li), 128-byte segments; inputs past K read zero
            const rsrc_t r = make_rsrc(a.xs0 + (size_t)t * K * B, (uint32_t)K * (uint32_t)B * 4u);
#pragma unroll
            for (int e = 0; e < WD_NKX; e++) vx[e] = buf_load(r, (uint32_t)g * 4u + (uint32_t)k0 * (uint32_t)B * 4u, (uint32_t)(e * 16) * (uint32_t)B * 4u);
        }
    };
    // x_t of this layer -> xt.  Above layer 0 it is the lower layer's h_t: wait for its four counters (steps published >= t + 1).
    // (Measured and not kept: polling both layers' counters in one round and requesting both tiles together -- the siblings' counters
    // become visible ~3 k cycles after a group's own publication, the lower layer's were there a stage ago: 17.0 against 15.4 k cycles per
    // step; a look at the own counters riding along with this tile's loads, to skip the second poll: 16.0 k.)
    auto stage_x = [&](int t) {
        if (l > 0) {
            if (!lost) {
                lost = !wd_wait8(rf, own_off, 0u, prev_off, (uint32_t)t + 1u, a.max_polls, a.err_local);
                if (lost) wd_lost(a.err, a.err_local);
            }
            const rsrc_t r = make_rsrc(a.hseq[l - 1] + (size_t)t * B * WD_H, step_bytes);      // rows past the batch read zero (range check)
#pragma unroll
            for (int e = 0; e < 8; e++) vx[e] = wd_load_coh(r, rm_off, (uint32_t)(e * 16 * 4));
#pragma unroll
            for (int e = 0; e < 8; e++) xt[srow * XS + sk + 16 * e] = lost ? __builtin_nanf("") : vx[e];
        } else if (a.xs0_btf && (K & 3) == 0) {
#pragma unroll
            for (int e = 0; e < WD_NKX / 4; e++)
                if (4 * (sk + 16 * e) < K) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++) xt[srow * XS + 4 * (sk + 16 * e) + c4] = vx[4 * e + c4];
                }
        } else if (a.xs0_btf) {
#pragma unroll
            for (int e = 0; e < WD_NKX; e++)
                if (sk + 16 * e < 2 * KPx) xt[srow * XS + sk + 16 * e] = vx[e];      // (masked loads: the pad column of an odd width is zero)
        } else {
#pragma unroll
            for (int e = 0; e < WD_NKX; e++)
                if (k0 + 16 * e < 2 * KPx) xt[li * XS + k0 + 16 * e] = k0 + 16 * e < K ? vx[e] : 0.f;      // (the pad column of an odd width: zero, every step -- the tile's LDS is reused)
        }
    };
    // h_{t-1}, all 128 units: wait for this layer's four counters (>= t), request the tile; it is written to LDS behind the input half's MFMAs
    float vh[8];
    auto h_issue = [&](int t) {
        if (!lost) {
            lost = !wd_wait8(rf, own_off, (uint32_t)t, prev_off, 0u, a.max_polls, a.err_local);
            if (lost) wd_lost(a.err, a.err_local);
        }
        const rsrc_t r = make_rsrc(a.hseq[l] + (size_t)(t - 1) * B * WD_H, step_bytes);
#pragma unroll
        for (int e = 0; e < 8; e++) vh[e] = wd_load_coh(r, rm_off, (uint32_t)(e * 16 * 4));
    };

    // the two accumulator elements this wave owns: e = 2 wave, 2 wave + 1 -> rows (e & 3) + 8 (e >> 2) + 4 lh of the tile
    const int orow = 2 * (wave & 1) + 8 * (wave >> 1) + 4 * lh;
    const int hidx = orow * WD_HS + q * 32 + li;
    const uint32_t out_off = (uint32_t)(((size_t)(row0 + orow) * WD_H + q * 32 + li) * 4);

    float ax[WD_NKX];
    auto read_ax = [&]() {
#pragma unroll
        for (int j = 0; j < WD_NKX; j++) ax[j] = j < nkx ? xt[li * XS + 2 * (xb + j) + lh] : 0.f;
    };
    if (l == 0) x0_issue(0);
    stage_x(0);
    __syncthreads();
    read_ax();
    OSL_TS_DECL
    for (int t = 0; t < T; t++) {
        OSL_TS(0)
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[g][e] = 0.f;
        // ---- gate GEMM, this wave's k-pairs.  r, z, gi_n <- x_t . W_i^T (x_t's fragments were read at the end of the previous step;
        // the tile of h_{t-1} is requested first and lands underneath) ----
        if (t > 0) h_issue(t);
#pragma unroll
        for (int j = 0; j < WD_NKX; j++) {
            if (j < nkx) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[j], wxr[j][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[j], wxr[j][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[j], wxr[j][2], acc[2], 0, 0, 0);
            }
        }
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 8; e++) ht[srow * WD_HS + sk + 16 * e] = lost ? __builtin_nanf("") : vh[e];
        }
        __syncthreads();
        OSL_TS(6)                                               // counters of this layer, h tile (under the input half's MFMAs), barrier
        // ---- r, z, gh_n <- h_{t-1} . W_h^T ----
        float hp0, hp1;
        {
            float ah[WD_NKH];
#pragma unroll
            for (int j = 0; j < WD_NKH; j++) ah[j] = ht[li * WD_HS + 2 * (hb + j) + lh];
            hp0 = ht[hidx]; hp1 = ht[hidx + WD_HS];            // h_{t-1} of the pair this wave updates
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hp0), "+v"(hp1) :: "memory");
            __syncthreads();                                    // every wave has its fragments: the tiles' LDS becomes the exchange buffer
            if (l == 0 && t + 1 < T) x0_issue(t + 1);           // (vx was consumed by stage_x(t))
#pragma unroll
            for (int j = 0; j < WD_NKH; j++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[j], whr[j][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[j], whr[j][1], acc[1], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[j], whr[j][2], acc[3], 0, 0, 0);
            }
        }
        OSL_TS(1)                                               // gate GEMM slice
        // ---- partial sums: all sixteen elements of every gate as 16-byte pieces (no per-element branches: the first form spent more
        // cycles issuing 56 conditional 4-byte writes than on the MFMAs); the owner of a pair reads its 8 bytes from the other seven ----
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                *reinterpret_cast<f32x4 *>(xch + (size_t)(((wave * 4 + g) * 4 + qd) * 64 + lane) * 4) =
                    (f32x4){acc[g][4 * qd], acc[g][4 * qd + 1], acc[g][4 * qd + 2], acc[g][4 * qd + 3]};
        __syncthreads();
        OSL_TS(2)                                               // partial sums -> LDS, barrier
        float own[4][2];
#pragma unroll
        for (int g = 0; g < 4; g++) { own[g][0] = 0.f; own[g][1] = 0.f; }
        {
            const int oq = wave >> 1, op = 2 * (wave & 1);                           // quad and position of element 2 wave
#pragma unroll
            for (int ww = 0; ww < 8; ww++)                                           // (fixed order 0..7, own partial included: deterministic)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const osk::f2 v = *reinterpret_cast<const osk::f2 *>(xch + (size_t)(((ww * 4 + g) * 4 + oq) * 64 + lane) * 4 + op);
                    own[g][0] += v[0]; own[g][1] += v[1];
                }
        }
        // ---- cell update of the pair (rows orow, orow + 1; unit q * 32 + li) ----
        const CellPair cp = gru_cell_pair((osk::f2){own[0][0], own[0][1]}, (osk::f2){own[1][0], own[1][1]}, (osk::f2){own[2][0], own[2][1]},
                                          (osk::f2){own[3][0], own[3][1]}, (osk::f2){hp0, hp1}, nb_r, nb_z, nb_n, b_hn);
        {
            const rsrc_t rh = make_rsrc(a.hseq[l] + (size_t)t * B * WD_H, step_bytes);          // rows past the batch are dropped (range check)
            wd_store_coh(rh, out_off, 0u, cp.hn[0]);
            wd_store_coh(rh, out_off, (uint32_t)WD_H * 4u, cp.hn[1]);
            if (SAVE) {
                const size_t so = (size_t)t * B * WD_H;
                const rsrc_t rr = make_rsrc(a.sv_r[l] + so, step_bytes), rz = make_rsrc(a.sv_z[l] + so, step_bytes),
                             rn = make_rsrc(a.sv_n[l] + so, step_bytes), rg = make_rsrc(a.sv_g[l] + so, step_bytes);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t so_i = (uint32_t)(i * WD_H * 4);
                    osk::buf_store_nt(rr, out_off, so_i, cp.r[i]); osk::buf_store_nt(rz, out_off, so_i, cp.z[i]);
                    osk::buf_store_nt(rn, out_off, so_i, cp.n[i]); osk::buf_store_nt(rg, out_off, so_i, cp.ghn[i]);
                }
            }
            if (t == T - 1 && a.h_last[l]) {
#pragma unroll
                for (int i = 0; i < 2; i++)
                    if (row0 + orow + i < B) a.h_last[l][(size_t)(q * 32 + li) * B + row0 + orow + i] = cp.hn[i];
            }
        }
        OSL_TS(3)                                               // sum of the partials, cell update, stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slice stores are acknowledged
        __syncthreads();                                        // ... every wave's; and ht / xch / xt are free again
        if (threadIdx.x == 0 && !(l == a.drop_layer && t >= a.drop_step))
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)t + 1u, rf, own_off + (uint32_t)q * 4u, 0u, WD_AUX);
        OSL_TS(4)                                               // store acknowledgements, barrier, counter
        if (t + 1 < T) {
            stage_x(t + 1);
            __syncthreads();
            read_ax();
            OSL_TS(5)                                           // the lower layer's counters, x tile -> LDS, barrier, its fragments
        }
    }
#ifdef OS_LAYER_TS
    if (tile == 0 && q == 0 && threadIdx.x == 0)
        printf("gru_wide_kernel layer %d K=%d T=%d cycles per step (wave 0): recurrent half %llu | partials + barrier %llu | sum + cell + stores %llu | acks + barrier + counter %llu | x wait + tile + fragments %llu | h wait + tile under the input half %llu | sum %llu\n",
               l, K, T, ts_sum[1] / T, ts_sum[2] / T, ts_sum[3] / T, ts_sum[4] / T, ts_sum[5] / T, ts_sum[6] / T,
               (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5] + ts_sum[6]) / T);
#endif
}

}  // namespace osg

using namespace osg;

static size_t wide_lds_bytes(int K0)
{
    const int KPx = (K0 + 1) / 2, XS = (2 * KPx + 1) > WD_HS ? 2 * KPx + 1 : WD_HS;
    const size_t tiles = (size_t)32 * (WD_HS + XS);
    return (tiles > (size_t)WD_XCH ? tiles : (size_t)WD_XCH) * sizeof(float);      // the exchange buffer aliases the tiles
}

// every workgroup resident at once, a tile's workgroups on one XCD: 4 n slots per tile, cu_count / 8 slots per XCD
bool os_gru_wide_eligible(os_ctx *ctx, int B, int T, int K0, int H, int n)
{
    if (ctx->tune_gru_wide == 0 || ctx->tune_gru_stack == 0 || H != WD_H || K0 > 192 || K0 < 1 || n < 1 || n > 8 || T < 1) return false;
    const int tiles = (B + 31) / 32, per_xcd = (tiles + 7) / 8 * 4 * n;
    if (per_xcd > ctx->cu_count / 8) return false;
    return (size_t)T * B * (K0 > H ? K0 : H) * 4 < ((size_t)1 << 31) && wide_lds_bytes(K0) <= (size_t)160 * 1024;
}

// One gru_wide_kernel launch over the n layers described by `a` (a.flags and the error fields are filled in here).
int os_gru_launch_wide(os_ctx *ctx, WideArgs &a, bool save, hipStream_t s)
{
    const size_t nfl = (size_t)a.n * a.tiles * 4;
    if (ctx->stack_flags_n < nfl) {
        if (ctx->stack_flags) OS_HIP(ctx, hipFree(ctx->stack_flags));
        ctx->stack_flags = nullptr; ctx->stack_flags_n = 0;
        OS_HIP(ctx, hipMalloc((void **)&ctx->stack_flags, nfl * sizeof(uint32_t)));
        ctx->stack_flags_n = nfl;
    }
    a.flags = ctx->stack_flags;
    a.err = ctx->stack_err_dev; a.err_local = ctx->stack_err_local; a.max_polls = ctx->stack_max_polls;
    a.drop_layer = ctx->stack_dbg_drop_layer; a.drop_step = ctx->stack_dbg_drop_step;
    OS_HIP(ctx, hipMemsetAsync(a.flags, 0, nfl * sizeof(uint32_t), s));
    if (!ctx->wide_attr_set) {
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->wide_attr_set = true;
    }
    const int slot = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, "gru_wide_kernel");
    const dim3 grid(8u * (unsigned)((a.tiles + 7) / 8 * 4 * a.n)), block(512);
    const size_t lds = wide_lds_bytes(a.K0);
    if (save) hipLaunchKernelGGL(gru_wide_kernel<true>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(gru_wide_kernel<false>, grid, block, lds, s, a);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return os_stack_verify(ctx, s, "gru_wide_kernel");
}
