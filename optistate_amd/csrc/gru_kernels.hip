// gru_kernels.hip -- GRU head of OptiState (reference gru/gru_model.py:7-49) on gfx950 matrix cores.
//
// One launch per layer runs the whole T-step recurrence for a tile of BM trajectories ("rows"):
//   gates[BM x 3H] = [x_t | h_{t-1}] . [W_ih | W_hh]^T      -> v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD)
//   r,z = sigmoid, n = tanh(gi_n + r*gh_n), h = (1-z)*n + z*h  -> VALU epilogue on the accumulator registers
// Layout decisions (MI355X-first, not a cuDNN translation):
//   * activations are structure-of-arrays [T][K][B] (trajectory index fastest), the same layout the Kalman kernels
//     stream, so an MFMA A-fragment (32 rows x 1 k) is one coalesced 128-byte segment per k straight from global/L2;
//   * weights are re-packed once (os_gru_load) into B-fragment order: for each 32-column chunk and each k-pair the 64
//     floats lane l needs (W[g*H + chunk*32 + (l&31)][2kp + (l>>5)]) are contiguous, so a wave streams its chunk's
//     weights with perfectly coalesced dword loads that stay L2-resident (<= 1.7 MB for the reference config);
//   * a wave owns one 32-column chunk of all three gates for RBW row blocks: the three gate pre-activations of a hidden
//     unit land in the same lane, so the cell update needs no cross-lane traffic;
//   * h lives in LDS ([BM][H+1] floats, odd stride -> conflict-free ds_read_b32 A-fragments), never in HBM; the layer's
//     output sequence is written back in SoA through a cooperative LDS->global pass (coalesced).
#include "launch.hpp"

#include <math.h>

#include <type_traits>
#include "gru_common.hpp"
#include "kf_device.hpp"   // buffer addressing helpers (make_rsrc / buf_load)
#include "gru_device.hpp"  // gru_cell_pair, LDS-DMA / inline-asm LDS helpers

namespace osg {

using osk::buf_load;
using osk::make_rsrc;
using osk::rsrc_t;

// One half of the gate GEMM (input part: XPART, accumulates gi_n into acc[.][2]; recurrent part: gh_n into acc[.][3]).
// Software pipeline, DEPTH k-pairs deep: the B fragments (coalesced dword loads of the fragment-ordered weights,
// L2-resident) and the A fragments (x part: one 128-B segment per k straight from the SoA stream; h part: LDS) of
// k-pair q+DEPTH are requested right after the MFMAs of k-pair q issue, so DEPTH * 3 * RBW MFMAs (>= 1500 cycles) cover
// each L2 round trip; a sched_barrier per k-pair keeps hipcc from hoisting every load to the top of the loop.
template <int RBW, bool XPART, int DEPTH = 4, typename AF>
__device__ __forceinline__ void mfma_part(f32x16 (*acc)[4], const float *wbase, int KP, int lane, AF afrag)
{
    float wb[DEPTH][3], ab[DEPTH][RBW];
    // weights: wave-uniform descriptor + SGPR offset (k-pair, gate) + constant per-lane offset -> no address VALU
    const rsrc_t w = make_rsrc(wbase, (uint32_t)KP * 3 * 256);
    const uint32_t wl = (uint32_t)lane * 4u;
    int q0 = 0;
    // Steady state: every k-pair of the block and every re-request is in range, so neither the prologue nor the body has a
    // conditional, and the loop is entered from that prologue ONLY.  (Round 4: hipcc's wait-count pass keeps one state per basic
    // block.  With the requests behind `if (q + DEPTH < KP)`, or with a guarded prologue merging into the loop header, the number
    // of loads younger than the fragment it needs differs per path, the merge keeps the smallest, and the pass drained the whole
    // queue -- s_waitcnt vmcnt(2) / (1) / (0) at the top of every block -- in every GRU layer kernel: the prefetch distance was
    // one k-pair, not DEPTH.  Like this it emits the exact counts, e.g. vmcnt(15) for DEPTH = 4, RBW = 2.)
    if (KP >= 2 * DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
#pragma unroll
            for (int g = 0; g < 3; g++) wb[j][g] = buf_load(w, wl, (uint32_t)(j * 3 + g) * 256u);
#pragma unroll
            for (int rb = 0; rb < RBW; rb++) ab[j][rb] = afrag(j, rb);
            // in request order: left alone, the scheduler issues slot 0's first fragment LAST, and the wait for the youngest
            // load at the loop header is vmcnt(0) on every iteration
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; q0 + 2 * DEPTH <= KP; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                const int q = q0 + j;
#pragma unroll
                for (int rb = 0; rb < RBW; rb++) {
                    const float av = ab[j][rb];
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][1], acc[rb][1], 0, 0, 0);
                    acc[rb][XPART ? 2 : 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][2], acc[rb][XPART ? 2 : 3], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; g++)
                    wb[j][g] = buf_load(w, wl, __builtin_amdgcn_readfirstlane((uint32_t)((q + DEPTH) * 3 + g) * 256u));
#pragma unroll
                for (int rb = 0; rb < RBW; rb++) ab[j][rb] = afrag(q + DEPTH, rb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            if (j < KP) {
#pragma unroll
                for (int g = 0; g < 3; g++) wb[j][g] = buf_load(w, wl, (uint32_t)(j * 3 + g) * 256u);
#pragma unroll
                for (int rb = 0; rb < RBW; rb++) ab[j][rb] = afrag(j, rb);
            }
        }
    }
    // tail: the last DEPTH .. 2 DEPTH - 1 k-pairs, requests guarded
    for (; q0 < KP; q0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const int q = q0 + j;
            if (q < KP) {
#pragma unroll
                for (int rb = 0; rb < RBW; rb++) {
                    const float av = ab[j][rb];
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][1], acc[rb][1], 0, 0, 0);
                    acc[rb][XPART ? 2 : 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[j][2], acc[rb][XPART ? 2 : 3], 0, 0, 0);
                }
                if (q + DEPTH < KP) {
#pragma unroll
                    for (int g = 0; g < 3; g++)
                        wb[j][g] = buf_load(w, wl, __builtin_amdgcn_readfirstlane((uint32_t)((q + DEPTH) * 3 + g) * 256u));
#pragma unroll
                    for (int rb = 0; rb < RBW; rb++) ab[j][rb] = afrag(q + DEPTH, rb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// RBW = row blocks (of 32 trajectories) per wave.  NCH = H/32 column chunks; with NCH < 4 several waves share a chunk
// and split the rows.  BM = 32 * RBW * max(1, 4/NCH).
// OCC = workgroups per CU the register budget is sized for.  Two resident workgroups matter more than a bigger row
// tile: while one sits in its cell update / barrier the other keeps the matrix pipe busy (measured, B = 65,536,
// T = 100: (60,128,4) 86 -> 114 TFLOP/s, (60,64,1) 82 -> 102).
// h is double-buffered in LDS ([2][BM][H+1]): step t reads h_{t-1} from one buffer and writes h_t into the other, so a
// step needs ONE barrier, and the SoA write-back of h_{t-1} is issued at the top of step t, under its MFMAs.
template <int RBW, int OCC>
__global__ __launch_bounds__(256, OCC) void gru_layer_kernel(const LayerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float hl2[];   // [2][BM][HS]
    // the wave index is wave-uniform but derived from threadIdx: readfirstlane makes that provable, so descriptors built
    // from it stay in SGPRs (otherwise hipcc wraps every buffer_load in a waterfall loop)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, HS = H + 1, NCH = H >> 5;
    const int WPC = NCH >= 4 ? 1 : 4 / NCH;          // waves per chunk
    const int BM = 32 * RBW * WPC;
    const int chunk = NCH >= 4 ? wave : wave % NCH;
    const int rgrp = NCH >= 4 ? 0 : wave / NCH;
    const int row_blk0 = rgrp * RBW;                  // first row block (within the tile) of this wave
    const int tile_row0 = blockIdx.x * BM;
    const int li = lane & 31, lh = lane >> 5;
    const size_t B = (size_t)a.B;

    for (int i = threadIdx.x; i < BM * HS; i += 256) hl2[i] = 0.f;   // h0 = 0 (gru/gru_model.py:27)

    const float *wx = a.w + (size_t)chunk * chunk_floats(a.KPx, a.KPh);
    const float *wh = wx + (size_t)a.KPx * 3 * 64;
    const float *bias = wh + (size_t)a.KPh * 3 * 64;
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];

    // A-fragment rows of this lane, clamped to the last trajectory (rows past B only feed their own, never-stored
    // outputs); xoff = byte offset of (row, k = lh) inside one step's [K][B] block: the k-pair is added as an SGPR offset
    const uint32_t rowB = (uint32_t)a.B * 4u;
    int growc[RBW];
    uint32_t xoff[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; rb++) {
        const int g = tile_row0 + (row_blk0 + rb) * 32 + li;
        growc[rb] = g < a.B ? g : a.B - 1;
        xoff[rb] = (uint32_t)growc[rb] * 4u + (uint32_t)lh * rowB;
    }
    __syncthreads();

    // cooperative, coalesced write-back of one h tile in SoA ([H][B] block at dst)
    auto write_back = [&](const float *hsrc, float *dst) {
        for (int i = threadIdx.x; i < BM * H; i += 256) {
            const int row = i % BM, k = i / BM;
            const int g = tile_row0 + row;
            if (g < a.B) dst[(size_t)k * B + g] = hsrc[row * HS + k];
        }
    };

    OSL_TS_DECL
    for (int t = 0; t < a.T; t++) {
        OSL_TS(0)
        const float *hl = hl2 + (t & 1) * BM * HS;          // h_{t-1}
        float *hn_buf = hl2 + ((t + 1) & 1) * BM * HS;       // h_t
        if (t > 0 && a.seq_out) write_back(hl, a.seq_out + (size_t)(t - 1) * H * B);
        OSL_TS(1)                                            // write-back of h_{t-1} (LDS -> SoA)

        f32x16 acc[RBW][4];
#pragma unroll
        for (int rb = 0; rb < RBW; rb++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rb][g][e] = 0.f;

        // ---- gates = [x_t | h_{t-1}] . [W_ih | W_hh]^T, software-pipelined DEPTH k-pairs deep (see mfma_part) ----
        const float *xt = a.xs + (size_t)t * a.K * B;
        const rsrc_t rx = make_rsrc(xt, (uint32_t)a.K * rowB);
        const int KPfull = a.K / 2;       // k-pairs with both k in range; an odd K leaves one masked tail pair
        mfma_part<RBW, true>(acc, wx, KPfull, lane, [&](int q, int rb) {
            return buf_load(rx, xoff[rb], __builtin_amdgcn_readfirstlane((uint32_t)(2 * q) * rowB));   // k = 2q + lh: see xoff
        });
        if (a.KPx > KPfull) {
            const int q = KPfull;
            const float w_r = wx[(q * 3 + 0) * 64 + lane], w_z = wx[(q * 3 + 1) * 64 + lane], w_n = wx[(q * 3 + 2) * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RBW; rb++) {
                const float av = (lh == 0) ? xt[(size_t)(2 * q) * B + growc[rb]] : 0.f;
                acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w_r, acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w_z, acc[rb][1], 0, 0, 0);
                acc[rb][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w_n, acc[rb][2], 0, 0, 0);
            }
        }
        OSL_TS(2)                                            // x half of the gate GEMM
        mfma_part<RBW, false>(acc, wh, a.KPh, lane, [&](int q, int rb) {
            return hl[((row_blk0 + rb) * 32 + li) * HS + 2 * q + lh];
        });
        OSL_TS(3)                                            // h half

        // ---- cell update on the accumulator layout: col = lane&31 (hidden unit), row = (e&3) + 8*(e>>2) + 4*(lane>>5)
        rsrc_t rs_r, rs_z, rs_n, rs_g, rs_h;
        if (a.sv_r) {
            const size_t so_t = (size_t)t * B * H;
            const uint32_t sbytes = (uint32_t)a.B * (uint32_t)H * 4u;
            rs_r = make_rsrc(a.sv_r + so_t, sbytes); rs_z = make_rsrc(a.sv_z + so_t, sbytes); rs_n = make_rsrc(a.sv_n + so_t, sbytes);
            rs_g = make_rsrc(a.sv_g + so_t, sbytes); rs_h = make_rsrc(a.sv_h + so_t, sbytes);
        }
        CellPair cp;
#pragma unroll
        for (int rb = 0; rb < RBW; rb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = (row_blk0 + rb) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int hidx = row * HS + chunk * 32 + li;
                // biases folded into the exp2 arguments: one FMA per gate instead of add + mul; the arithmetic runs on element
                // pairs (gru_cell_pair): computed at the even element, picked up at the odd one
                if ((e & 1) == 0) {
                    const int hidx1 = hidx + HS;                        // element e + 1: the next row
                    cp = gru_cell_pair((osk::f2){acc[rb][0][e], acc[rb][0][e + 1]}, (osk::f2){acc[rb][1][e], acc[rb][1][e + 1]},
                                       (osk::f2){acc[rb][2][e], acc[rb][2][e + 1]}, (osk::f2){acc[rb][3][e], acc[rb][3][e + 1]},
                                       (osk::f2){hl[hidx], hl[hidx1]}, nb_r, nb_z, nb_n, b_hn);
                }
                const float r = cp.r[e & 1], z = cp.z[e & 1], n = cp.n[e & 1], ghn = cp.ghn[e & 1], hn = cp.hn[e & 1];
                hn_buf[hidx] = hn;
                if (a.sv_r) {
                    // write-once streams (non-temporal: they must not evict the L2-resident weights) through the step's [B][H]
                    // descriptors: one per-lane offset per row block, the element's row in a wave-uniform offset, rows past the batch
                    // dropped by the range check (flat addresses cost two registers per store in flight: 21 spilled VGPRs)
                    const uint32_t svo = (uint32_t)(((size_t)(tile_row0 + (row_blk0 + rb) * 32 + 4 * lh) * H + chunk * 32 + li) * 4);
                    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)((e & 3) + 8 * (e >> 2)) * (uint32_t)H * 4u);
                    osk::buf_store_nt(rs_r, svo, so, r); osk::buf_store_nt(rs_z, svo, so, z); osk::buf_store_nt(rs_n, svo, so, n);
                    osk::buf_store_nt(rs_g, svo, so, ghn); osk::buf_store_nt(rs_h, svo, so, hn);
                }
            }
        }
        OSL_TS(4)                                            // cell update
        lds_barrier();   // h_t complete; every wave is also done with h_{t-1}, which the NEXT epilogue overwrites
        OSL_TS(5)
    }
#ifdef OS_LAYER_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("gru_layer_kernel<%d,%d> K=%d cycles per step: write-back %llu | x half %llu | h half %llu | cell update %llu | barrier %llu | sum %llu\n",
               RBW, OCC, a.K, ts_sum[1] / a.T, ts_sum[2] / a.T, ts_sum[3] / a.T, ts_sum[4] / a.T, ts_sum[5] / a.T,
               (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5]) / a.T);
#endif
    const float *hT = hl2 + (a.T & 1) * BM * HS;
    if (a.seq_out) write_back(hT, a.seq_out + (size_t)(a.T - 1) * H * B);
    if (a.h_last) write_back(hT, a.h_last);
}

// ---------------------------------------------------------------------------------------------------------------
// gru_layer_stage_kernel -- large batches, H = 128, inference: the x tile of a step goes global -> LDS by LDS-DMA one step
// ahead, so the MFMA loop reads BOTH halves of its A operand from LDS and the only vector loads it waits for are the
// L2-resident weight fragments.
//
// Why (round 4; in-kernel timestamps: profiles/r04_layer_timestamps.md): gru_layer_kernel fetches the x fragments straight
// from the SoA stream inside the pipelined k loop.  At B = 65,536 that stream (3.3 GB per layer and pass) comes from HBM,
// a wave's vector loads return in order, and so every HBM-latency x fragment holds up the weight fragments queued behind
// it -- and all four waves of a workgroup (one per 32-column chunk) fetch the same x tile again (5.8x the algorithmic
// bytes).  gru_layer_split_kernel cured the same disease at small batches by staging through registers; here
// the tile is 64 rows x K <= 188 inputs = 48 KB, too much for registers, so it travels by `buffer_load ... lds`
// (16 bytes per lane: one instruction moves 4 inputs x 64 rows), fetched ONCE per workgroup.
//
// LDS (80,896 B at K = 188: two workgroups per CU): hS [64][129] (odd stride: conflict-free A fragments) | xS [K4][64]
// (input-major, K4 = K rounded up to 4: exactly what the DMA writes, and lane li of an A fragment reads bank li).
// Single-buffered, two barriers per step:
//   MFMAs of step t (x part from xS, h part from hS)  | barrier 1: every wave is done reading xS and hS
//   DMA x_{t+1} -> xS (asynchronous), cell update of step t (h_{t-1} of the wave's own 2 x 16 elements lives in registers:
//   no LDS read), h_t -> hS and, as four 16-byte stores per row block, straight to seq_out in SoA (an accumulator's
//   elements e = 4j .. 4j+3 are four consecutive trajectories of one hidden unit) | vmcnt: the DMA has landed | barrier 2
// Every LDS access inside the T loop is inline assembly: hipcc treats an LDS-DMA as aliasing every later LDS access and
// puts s_waitcnt vmcnt(0) in front of each (which would serialise the DMA with the cell update and wait for the write-once
// seq_out stores as well).  An A fragment's register is re-requested behind the FIRST MFMA of the following k-pair: the
// matrix pipe reads an MFMA's A/B registers when the instruction starts, not when it issues.
// Bit-identical to gru_layer_kernel: same products, same accumulation order.
// ---------------------------------------------------------------------------------------------------------------

constexpr int STAGE_BM = 64, STAGE_DEPTH = 4;       // rows per wave pair of row blocks; k-pairs in flight

// One half of the gate GEMM with the A fragments in LDS.  ap: byte address of this lane's fragment element of k-pair 0, row
// block 0; QS / RS: byte strides of a k-pair / of the second row block (immediates).  XPART: gate n accumulates into acc[.][2]
// (gi_n), else acc[.][3] (gh_n).  Software pipeline STAGE_DEPTH k-pairs deep; slot j - 1 is re-requested behind the first
// MFMA of slot j.
template <bool XPART, int QS, int RS>
__device__ __forceinline__ void mfma_part_lds(f32x16 (*acc)[4], const float *wbase, int KP, int lane, uint32_t ap)
{
    constexpr int D = STAGE_DEPTH;
    constexpr int G = XPART ? 2 : 3;
    float wb[D][3], ab[D][2];
    const rsrc_t w = make_rsrc(wbase, (uint32_t)KP * 3 * 256);
    const uint32_t wl = (uint32_t)lane * 4u;
    uint32_t wo = 0;                                     // byte offset of k-pair q0's fragments (SGPR)
    // request k-pair (q0 + off_kp)'s fragments into slot j: two A elements from LDS (ap points at k-pair q0), three B fragments
#define OSL_REQ(j, off_kp)                                                                                             \
    {                                                                                                                   \
        ab[j][0] = lds_read_asm<(off_kp) * QS>(ap);                                                                     \
        ab[j][1] = lds_read_asm<(off_kp) * QS + RS>(ap);                                                                \
        wb[j][0] = buf_load(w, wl, wo + (uint32_t)((off_kp) * 3 + 0) * 256u);                                           \
        wb[j][1] = buf_load(w, wl, wo + (uint32_t)((off_kp) * 3 + 1) * 256u);                                           \
        wb[j][2] = buf_load(w, wl, wo + (uint32_t)((off_kp) * 3 + 2) * 256u);                                           \
    }
    // k-pair in slot j: wait for its A elements (the asm operands tie the wait to the registers: the MFMAs cannot move above it),
    // first MFMA, then -- behind a started MFMA of this k-pair, so that every MFMA of the previous one has read its operands --
    // re-request the previous k-pair's slot, then the other five MFMAs
#define OSL_STEP(j, jm, off_next, REQ)                                                                                 \
    {                                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ab[j][0]), "+v"(ab[j][1])::"memory");                                \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][0], wb[j][0], acc[0][0], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (REQ) OSL_REQ(jm, off_next)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][0], wb[j][1], acc[0][1], 0, 0, 0);                       \
        acc[0][G] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][0], wb[j][2], acc[0][G], 0, 0, 0);                       \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][1], wb[j][0], acc[1][0], 0, 0, 0);                       \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][1], wb[j][1], acc[1][1], 0, 0, 0);                       \
        acc[1][G] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[j][1], wb[j][2], acc[1][G], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
    // KP >= 4 and KP even (H = 128: KPh = 64; K even): prologue fills the four slots
    OSL_REQ(0, 0) OSL_REQ(1, 1) OSL_REQ(2, 2) OSL_REQ(3, 3)
    // steady state: NO conditional inside -- with the requests behind `if (q + D < KP)` hipcc's wait-count pass cannot tell how
    // many loads are younger than the one it needs (the count differs per path) and drains the queue (vmcnt(0)) at the top of
    // every block of four k-pairs: the prefetch distance collapses from four k-pairs to one (gru_layer_kernel's loop does that)
    int q0 = 0;
    for (; q0 + 2 * D <= KP; q0 += D) {
        // offsets relative to ap / wo (k-pair q0): slot j - 1 takes k-pair q0 + j - 1 + D
        OSL_STEP(0, 3, D - 1, q0 > 0)
        OSL_STEP(1, 0, D, true)
        OSL_STEP(2, 1, D + 1, true)
        OSL_STEP(3, 2, D + 2, true)
        ap += D * QS;
        wo += D * 3 * 256;
    }
    // tail: fewer than 2 D k-pairs left, every one of them already requested except those the guarded requests below add
    for (; q0 < KP; q0 += D) {
#define OSL_TAIL(j, jm, off_next)                                                                                      \
        if (q0 + (j) < KP) OSL_STEP(j, jm, off_next, (q0 + (j) >= 1 && q0 + (j) - 1 + D < KP))
        OSL_TAIL(0, 3, D - 1)
        OSL_TAIL(1, 0, D)
        OSL_TAIL(2, 1, D + 1)
        OSL_TAIL(3, 2, D + 2)
#undef OSL_TAIL
        ap += D * QS;
        wo += D * 3 * 256;
    }
#undef OSL_STEP
#undef OSL_REQ
}

// NCH = H / 32 column chunks: 4 (H = 128: a wave per chunk, 64-row tile) or 2 (H = 64: two waves per chunk on different
// halves of a 128-row tile, as in gru_layer_kernel).
template <int NCH>
__global__ __launch_bounds__(256, 2) void gru_layer_stage_kernel(const LayerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // hS [BM][H + 1] | xS [K4][BM]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int H = 32 * NCH, HS = H + 1, BM = STAGE_BM * (4 / NCH);
    constexpr int LPK = BM / 4, KPI = 64 / LPK;                   // DMA: lanes per input row of the tile, inputs per instruction
    const int chunk = wave % NCH, rgrp = wave / NCH;              // rgrp: which 64 rows of the tile
    const int tile_row0 = blockIdx.x * BM;
    const int li = lane & 31, lh = lane >> 5;
    const int K4 = (a.K + 3) & ~3;
    float *hS = sm, *xS = sm + BM * HS;                           // BM * HS floats: a multiple of 16 bytes in both shapes
    static_assert((BM * HS * 4) % 16 == 0, "xS must stay 16-byte aligned");
    for (int i = threadIdx.x; i < BM * HS; i += 256) hS[i] = 0.f;          // h0 = 0 (gru/gru_model.py:27)

    const float *wx = a.w + (size_t)chunk * chunk_floats(a.KPx, a.KPh);
    const float *wh = wx + (size_t)a.KPx * 3 * 64;
    const float *bias = wh + (size_t)a.KPh * 3 * 64;
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];
    const uint32_t rowB = (uint32_t)a.B * 4u;

    // x tile DMA: instruction j moves inputs KPI j .. KPI j + KPI - 1 of the BM rows (lane l: input KPI j + l / LPK, rows
    // 4 (l % LPK) .. + 3) = 256 consecutive floats of the input-major tile
    const uint32_t dvoff = (uint32_t)(lane / LPK) * rowB + (uint32_t)(tile_row0 + 4 * (lane % LPK)) * 4u;
    auto stage_x = [&](int t) {
        const rsrc_t rx = make_rsrc(a.xs + (size_t)t * a.K * a.B, (uint32_t)a.K * rowB);      // inputs past K read as zero (range check)
        for (int j = wave; j < K4 / KPI; j += 4)
            stage_dma16(rx, xS + j * 256, dvoff, __builtin_amdgcn_readfirstlane((uint32_t)(KPI * j) * rowB));
    };
    // LDS byte addresses of this lane's A-fragment elements (k-pair 0, row block 0) and of its 2 x 16 cell elements
    const uint32_t xS_b = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)xS;
    const uint32_t hS_b = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)hS;
    const uint32_t ax0 = xS_b + (uint32_t)(lh * BM + rgrp * 64 + li) * 4u;
    const uint32_t ah0 = hS_b + (uint32_t)((rgrp * 64 + li) * HS + lh) * 4u;
    const uint32_t hw0 = hS_b + (uint32_t)((rgrp * 64 + 4 * lh) * HS + chunk * 32 + li) * 4u;   // row (e & 3) + 8 (e >> 2) + 4 lh, col chunk * 32 + li

    stage_x(0);
    float hv[2][16];                                                  // h_{t-1} of the wave's own elements
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int e = 0; e < 16; e++) hv[rb][e] = 0.f;
    // seq_out / h_last stores: byte offset of (hidden unit chunk * 32 + li, trajectory tile_row0 + 4 lh) inside one [H][B] block;
    // rows past B aim past the descriptor's range (dropped)
    uint32_t so[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int row0 = tile_row0 + rgrp * 64 + rb * 32 + 4 * lh;
        so[rb] = (uint32_t)(chunk * 32 + li) * rowB + (uint32_t)row0 * 4u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    OSL_TS_DECL

    for (int t = 0; t < a.T; t++) {
        OSL_TS(0)
        f32x16 acc[2][4];
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rb][g][e] = 0.f;
        mfma_part_lds<true, 2 * BM * 4, 128>(acc, wx, a.KPx, lane, ax0);
        OSL_TS(1)                                        // x half of the gate GEMM
        mfma_part_lds<false, 8, 32 * HS * 4>(acc, wh, a.KPh, lane, ah0);
        OSL_TS(2)                                        // h half
        lds_barrier();                                   // barrier 1: xS and hS are free
        OSL_TS(3)
        if (t + 1 < a.T) stage_x(t + 1);
        // ---- cell update on the accumulator layout: col = lane & 31 (hidden unit), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
        const rsrc_t rs = make_rsrc(a.seq_out ? a.seq_out + (size_t)t * H * a.B : nullptr, a.seq_out ? (uint32_t)H * rowB : 0u);
        const rsrc_t rl = make_rsrc((t == a.T - 1 && a.h_last) ? a.h_last : nullptr, (t == a.T - 1 && a.h_last) ? (uint32_t)H * rowB : 0u);
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            // element pairs (2 p, 2 p + 1) sit in adjacent accumulator registers (gru_cell_pair: packed arithmetic, bit-identical
            // to the scalar form)
            using osk::f2;
#pragma unroll
            for (int pr = 0; pr < 8; pr++) {
                const int e0 = 2 * pr, e1 = e0 + 1;
                const CellPair cp = gru_cell_pair((f2){acc[rb][0][e0], acc[rb][0][e1]}, (f2){acc[rb][1][e0], acc[rb][1][e1]},
                                                  (f2){acc[rb][2][e0], acc[rb][2][e1]}, (f2){acc[rb][3][e0], acc[rb][3][e1]},
                                                  (f2){hv[rb][e0], hv[rb][e1]}, nb_r, nb_z, nb_n, b_hn);
                const f2 hn = cp.hn;
                hv[rb][e0] = hn[0]; hv[rb][e1] = hn[1];
                lds_write_asm<0>(hw0 + (uint32_t)((rb * 32 + (e0 & 3) + 8 * (e0 >> 2)) * HS) * 4u, hn[0]);   // (constant: folds into the offset field)
                lds_write_asm<0>(hw0 + (uint32_t)((rb * 32 + (e1 & 3) + 8 * (e1 >> 2)) * HS) * 4u, hn[1]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 v = {hv[rb][4 * j], hv[rb][4 * j + 1], hv[rb][4 * j + 2], hv[rb][4 * j + 3]};
                const uint32_t off = (tile_row0 + rgrp * 64 + rb * 32 + 8 * j + 4 * lh < a.B) ? so[rb] + (uint32_t)(8 * j) * 4u : 0x80000000u;
                buf_store4(rs, off, 0, v);
                buf_store4(rl, off, 0, v);
            }
        }
        OSL_TS(4)                                        // DMA issue + cell update + stores
        // the DMA (older than this step's 16 store instructions) has landed; the stores themselves may still be in flight
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        lds_barrier();                                   // barrier 2: h_t and x_{t+1} are in LDS
        OSL_TS(5)
    }
#ifdef OS_LAYER_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("gru_layer_stage_kernel<%d> K=%d cycles per step: x half %llu | h half %llu | barrier 1 %llu | DMA issue + cell + stores %llu | DMA wait + barrier 2 %llu | sum %llu\n",
               NCH, a.K, ts_sum[1] / a.T, ts_sum[2] / a.T, ts_sum[3] / a.T, ts_sum[4] / a.T, ts_sum[5] / a.T,
               (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5]) / a.T);
#endif
}

// Small-batch variant: EIGHT waves per workgroup on one 32-row tile.  The concatenated k-pair range [x part | h part] of
// the gate GEMM is cut into PARTS = 8 / NCH slices per 32-column chunk (H = 128: input half / recurrent half; H = 64: four
// slices); wave (chunk, part) accumulates its slice, parts 1.. hand their partial sums to part 0 through
// LDS, part 0 adds them up and does the cell update of the chunk.  With B <= 8 k a 32-row tile is all a CU gets (the
// recurrence cannot be split across workgroups), so a second wave per SIMD is the only thing that can hide a wave's L2 /
// LDS stalls -- the same lever as the eight-wave backward sweep.  Numerically identical to gru_layer_kernel up to the order
// of PARTS - 1 additions per gate.
// STACK (gru_stack_kernel below): the layers of a small batch run as ONE launch, blockIdx.y = layer, and layer l consumes
// layer l - 1's sequence step by step through per-(layer, tile) progress flags in global memory instead of waiting for the
// whole kernel: `prev` = the producer's flag (null for the first layer of the launch), `mine` = this workgroup's.
struct StackSync {
    const uint32_t *prev;
    uint32_t *mine;
    int32_t *err = nullptr;          // the context's error word (pinned host memory, device-mapped): bit 0 = a wait expired
    int32_t *err_local = nullptr;    // its twin in device memory (adam_kernel polls this one)
    uint32_t max_polls = 1u << 22;
    int drop_from = -1;              // tests (OS_STACK_DBG_DROP): this workgroup stops publishing from this step on
};
// sequence elements that cross workgroups inside one launch: sc0 | sc1 accesses (coherent across the XCDs' L2s)
__device__ __forceinline__ float buf_load_coh(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 17));
}
// wait until the producer has published step `need - 1`; false after ~4 M polls (seconds): the caller then reports it in the
// context's error word (stack_lost), latches the loss (no further waits in this launch) and poisons its input with NaN, so a lost
// producer ends in an error return of the call (or of the next one: launch.hpp) and NaN outputs -- never in a hung GPU and never
// in plausible wrong numbers
__device__ __forceinline__ void stack_lost(int32_t *err, int32_t *err_local)
{
    if ((threadIdx.x & 63) == 0) {
        if (err_local) __hip_atomic_fetch_or(err_local, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err) __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ bool stack_wait(const uint32_t *flag, uint32_t need, uint32_t max_polls = 1u << 22, const int32_t *err_local = nullptr)
{
    for (uint32_t spin = 0; spin < max_polls; spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if ((spin & 1023u) == 1023u && stack_lost_already(err_local)) return false;      // somebody else gave up: so do we
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

// SAVE: the training forward (r, z, n, gh_n, h_t saved row-major).  A template parameter since round 5: with the five output
// streams' address arithmetic compiled in, the H = 64 inference instantiations spilled 82 VGPRs.
// GI (os_gru_forward_windows): the input half of the gate GEMM is not computed here -- the launch's trajectories are the sliding
// windows of ONE row stream (window b = rows b .. b + T - 1, gru/gru_test.py:138-140), so x_t W_ih^T of a row is shared by the T
// windows that contain it and comes precomputed (gru_gi_kernel) as a.gi; part 0 adds its tile (one step ahead, in registers)
// to the recurrent half's accumulators.  a.K = a.KPx = 0 in that mode: no x tile, the whole k range is the recurrent half.
template <int NCH, bool STACK, bool SAVE, bool GI = false>
__device__ __forceinline__ void split_layer_body(const LayerArgs &a, const StackSync sy)
{
    constexpr int PARTS = 8 / NCH;
#ifdef OS_LAYER_TS
    const unsigned long long ts_start = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][32][HS] h double buffer | [2][32][XS] x tiles | [NCH][PARTS-1][64][64] exchange
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, HS = H + 1;
    constexpr int BM = 32;
    const int chunk = wave % NCH, part = wave / NCH;               // part 0 also does the chunk's cell update
    const int tile_row0 = blockIdx.x * BM;
    const int li = lane & 31, lh = lane >> 5;
    const size_t B = (size_t)a.B;
    // The x tile of a step is staged in LDS one step ahead (row-major [32][XS], XS = 2 KPx + 1: odd stride, and the pad
    // column of an odd input width stays zero).  Fetching the A fragments of the x half straight from the SoA stream inside
    // the MFMA loop ran that half at 0.112 ms per layer against 0.05 ms for its MFMAs: vector loads return in order per wave,
    // so every HBM-latency x fragment held up the L2-resident weight fragments queued behind it.
    const int XS = 2 * a.KPx + 1;
    float *hl2 = smem;
    float *xl2 = smem + 2 * BM * HS;
    float *xch = xl2 + 2 * BM * XS + (size_t)chunk * (PARTS - 1) * 64 * 64;

    for (int i = threadIdx.x; i < BM * HS; i += 512) hl2[i] = 0.f;   // h0 = 0 (gru/gru_model.py:27)
    for (int i = threadIdx.x; i < 2 * BM * XS; i += 512) xl2[i] = 0.f;

    const float *wx = a.w + (size_t)chunk * chunk_floats(a.KPx, a.KPh);
    const float *wh = wx + (size_t)a.KPx * 3 * 64;
    const float *bias = wh + (size_t)a.KPh * 3 * 64;
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];
    const uint32_t rowB = (uint32_t)a.B * 4u;
    const int g0 = tile_row0 + li;
    const int growc = g0 < a.B ? g0 : a.B - 1;
    // this part's slice of the k-pair range: x k-pairs [xb, xe), h k-pairs [hb, he)
    const int KPfull = GI ? 0 : a.KPx, total = KPfull + a.KPh;      // (GI: no x part; the weight image still has one, a.KPx locates the h part)
    const int qb = (int)((long)total * part / PARTS), qe = (int)((long)total * (part + 1) / PARTS);
    const int xb = qb < KPfull ? qb : KPfull, xe = qe < KPfull ? qe : KPfull;
    const int hb = (qb > KPfull ? qb : KPfull) - KPfull, he = (qe > KPfull ? qe : KPfull) - KPfull;
    // x tile staging: thread -> (input k0 + 16 e, row li), 128-byte segments of the [K][B] stream; inputs past K read zero
    // through the descriptor's range check (and are not written to the tile)
    constexpr int XE = 12;                                   // 16 inputs per pass: K <= 192
    const int xk0 = threadIdx.x >> 5;
    const uint32_t xsoff = (uint32_t)growc * 4u + (uint32_t)xk0 * rowB;
    float xr[XE];
    const bool consume = STACK && sy.prev != nullptr;               // x comes from a producer inside this launch
    bool lost = STACK && stack_lost_already(sy.err_local);          // latched: after one expired wait (anywhere in the launch) this workgroup stops waiting
    auto xfetch = [&](int t) {
        if (consume && !lost) {
            lost = !stack_wait(sy.prev, (uint32_t)t + 1u, sy.max_polls, sy.err_local);   // every lane polls (one request per wave): step t is published
            if (lost) stack_lost(sy.err, sy.err_local);
        }
        const rsrc_t rx = make_rsrc(a.xs + (size_t)t * a.K * B, (uint32_t)a.K * rowB);
#pragma unroll
        for (int e = 0; e < XE; e++) {
            if (e * 16 >= a.K) break;
            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(e * 16) * rowB);
            xr[e] = consume ? buf_load_coh(rx, xsoff, so) : buf_load(rx, xsoff, so);
            if (STACK && lost) xr[e] = __builtin_nanf("");
        }
    };
    auto xstage = [&](float *xl) {
#pragma unroll
        for (int e = 0; e < XE; e++) {
            if (e * 16 >= a.K) break;
            if (xk0 + e * 16 < a.K) xl[li * XS + xk0 + e * 16] = xr[e];
        }
    };
    __syncthreads();
    if (!GI) {
        xfetch(0);
        xstage(xl2);
    }
    __syncthreads();

    auto write_back = [&](const float *hsrc, float *dst) {
        for (int i = threadIdx.x; i < BM * H; i += 512) {
            const int row = i % BM, k = i / BM;
            const int g = tile_row0 + row;
            if (g < a.B) dst[(size_t)k * B + g] = hsrc[row * HS + k];
        }
    };
    // STACK: h_t -> seq_out[t] with write-through (sc0 sc1) stores right after the step, then the progress flag as an agent-scope
    // release once every wave's stores are acknowledged
    auto publish = [&](const float *hsrc, int t) {
        const rsrc_t rq = make_rsrc(a.seq_out + (size_t)t * H * B, (uint32_t)H * rowB);
        for (int i = threadIdx.x; i < BM * H; i += 512) {
            const int row = i % BM, k = i / BM;
            const int g = tile_row0 + row;
            if (g < a.B)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, hsrc[row * HS + k]), rq, (uint32_t)g * 4u, (uint32_t)k * rowB, 17);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are acknowledged (a workgroup-scope barrier does not wait for them)
        __syncthreads();
        if (threadIdx.x == 0 && !(sy.drop_from >= 0 && t >= sy.drop_from))
            __hip_atomic_store(sy.mine, (uint32_t)t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };

#ifdef OS_LAYER_TS
    const unsigned long long ts_pro = __builtin_readcyclecounter() - ts_start;
#endif
    const uint32_t svoff = (uint32_t)(((size_t)(tile_row0 + 4 * lh) * H + chunk * 32 + li) * 4);      // saved activations: row 4 lh of the tile, this lane's unit
    // GI: this lane's 48 entries of the gi tile of the step AFTER the one whose MFMAs are running (element e = window row
    // (e & 3) + 8 (e >> 2) + 4 lh of the tile, gate g): 128-byte segments of gi[row][g H + unit]; rows past the stream read zero
    // (PARTS == 2: each of a chunk's two waves owns eight of the sixteen elements -- see the exchange below -- and fetches those)
    constexpr bool HALVES = PARTS == 2;
    constexpr int NE = HALVES ? 8 : 16;
    float gn[3][NE];
    const rsrc_t rgi = make_rsrc(GI ? a.gi : nullptr, GI ? (uint32_t)a.gi_rows * 3u * (uint32_t)H * 4u : 0u);
    uint32_t gvo = (uint32_t)(((size_t)(tile_row0 + 4 * lh) * 3 * H + chunk * 32 + li) * 4);
    auto gfetch = [&](auto E0) {
        constexpr int e0 = decltype(E0)::value;
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int i = 0; i < NE; i++)
                gn[g][i] = buf_load(rgi, gvo, (uint32_t)((((e0 + i) & 3) + 8 * ((e0 + i) >> 2)) * 3 * H + g * H) * 4u);
        gvo += 3u * (uint32_t)H * 4u;                   // the next step's rows are one row further down
    };
    // (PARTS == 4, H = 64: 48 more registers across the MFMA slice do not fit -- 9 spilled VGPRs -- so part 0 requests its tile BEHIND
    // its slice instead, under the exchange)
    if (GI && HALVES) {
        if (part == 0) gfetch(std::integral_constant<int, 0>{});
        else gfetch(std::integral_constant<int, 8>{});
    }
    OSL_TS_DECL
    for (int t = 0; t < a.T; t++) {
        OSL_TS(0)
        const float *hl = hl2 + (t & 1) * BM * HS;          // h_{t-1}
        float *hn_buf = hl2 + ((t + 1) & 1) * BM * HS;       // h_t
        if (!STACK && t > 0 && a.seq_out) write_back(hl, a.seq_out + (size_t)(t - 1) * H * B);
        OSL_TS(1)                                            // write-back of h_{t-1} (LDS -> SoA)

        f32x16 acc[1][4];
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[0][g][e] = 0.f;

        const float *xl = xl2 + (t & 1) * BM * XS;          // x_t
        if (xe > xb)
            mfma_part<1, true, 8>(acc, wx + (size_t)xb * 3 * 64, xe - xb, lane, [&](int q, int) { return xl[li * XS + 2 * (q + xb) + lh]; });
        if (he > hb)
            mfma_part<1, false, GI ? 4 : 8>(acc, wh + (size_t)hb * 3 * 64, he - hb, lane, [&](int q, int) { return hl[li * HS + 2 * (q + hb) + lh]; });      // (GI: 48 registers hold the next gi tile)
        OSL_TS(2)                                            // this wave's slice of the gate GEMM
        // x_{t+1}: requested behind the last weight fragment, lands underneath the exchange and the cell update
        if (!GI && t + 1 < a.T) xfetch(t + 1);
        if (GI && !HALVES && part == 0) gfetch(std::integral_constant<int, 0>{});
        // Exchange of the partial sums.  PARTS == 2 (H = 128): the two waves of a chunk SWAP halves -- part 0 hands elements 8..15 to
        // part 1 and takes elements 0..7 from it -- and each does the cell update of its eight elements (rows 0-3, 8-11 / 16-19, 24-27
        // of the tile, + 4 lh): the same 64 values cross LDS as before, but the cell update, which was a third of a step on ONE wave
        // per SIMD while the other sat at the barrier, runs on both (round 5).  PARTS == 4 (H = 64): parts 1.. hand everything to part 0.
        if (HALVES) {
            float *dst = xch;
            if (part == 0) {
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 8; e < 16; e++) dst[(g * 16 + e) * 64 + lane] = acc[0][g][e];
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 8; e++) dst[(g * 16 + e) * 64 + lane] = acc[0][g][e];
            }
        } else if (part > 0) {
            float *dst = xch + (size_t)(part - 1) * 64 * 64;
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) dst[(g * 16 + e) * 64 + lane] = acc[0][g][e];
        }
        lds_barrier();   // partial sums in place; every wave is done reading h_{t-1}'s A fragments
        OSL_TS(3)                                            // x request, partial sums -> LDS, barrier
        // sum of the partials + cell update of elements [e0, e0 + ne) on the calling wave
        auto cell = [&](auto E0, auto NEL) {
            constexpr int e0 = decltype(E0)::value, ne = decltype(NEL)::value;
#pragma unroll
            for (int p = 0; p < (HALVES ? 1 : PARTS - 1); p++) {
                const float *src = xch + (size_t)p * 64 * 64;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = e0; e < e0 + ne; e++) acc[0][g][e] += src[(g * 16 + e) * 64 + lane];
            }
            rsrc_t rs_r, rs_z, rs_n, rs_g, rs_h;
            if (SAVE) {
                const size_t so_t = (size_t)t * B * H;
                const uint32_t sbytes = (uint32_t)a.B * (uint32_t)H * 4u;
                rs_r = make_rsrc(a.sv_r + so_t, sbytes); rs_z = make_rsrc(a.sv_z + so_t, sbytes); rs_n = make_rsrc(a.sv_n + so_t, sbytes);
                rs_g = make_rsrc(a.sv_g + so_t, sbytes); rs_h = make_rsrc(a.sv_h + so_t, sbytes);
            }
            if (GI) {
#pragma unroll
                for (int g = 0; g < 3; g++)
#pragma unroll
                    for (int i = 0; i < ne; i++) acc[0][g][e0 + i] += gn[g][i];      // r, z, n_x: the input half, shared by T windows
            }
            CellPair cp;
#pragma unroll
            for (int e = e0; e < e0 + ne; e++) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int hidx = row * HS + chunk * 32 + li;
                if ((e & 1) == 0)                                    // element pairs (gru_cell_pair): e + 1 is the next row
                    cp = gru_cell_pair((osk::f2){acc[0][0][e], acc[0][0][e + 1]}, (osk::f2){acc[0][1][e], acc[0][1][e + 1]},
                                       (osk::f2){acc[0][2][e], acc[0][2][e + 1]}, (osk::f2){acc[0][3][e], acc[0][3][e + 1]},
                                       (osk::f2){hl[hidx], hl[hidx + HS]}, nb_r, nb_z, nb_n, b_hn);
                const float r = cp.r[e & 1], z = cp.z[e & 1], n = cp.n[e & 1], ghn = cp.ghn[e & 1], hn = cp.hn[e & 1];
                hn_buf[hidx] = hn;
                if (SAVE) {
                    // one per-lane offset register for the 80 stores of a step, the element's row in a wave-uniform offset; rows
                    // past the batch fall outside the step's [B][H] descriptor and are dropped by the range check (flat
                    // addresses: two registers per store in flight -- the H = 64 instantiation spilled 82 VGPRs)
                    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)((e & 3) + 8 * (e >> 2)) * (uint32_t)H * 4u);
                    osk::buf_store_nt(rs_r, svoff, so, r); osk::buf_store_nt(rs_z, svoff, so, z); osk::buf_store_nt(rs_n, svoff, so, n);
                    osk::buf_store_nt(rs_g, svoff, so, ghn); osk::buf_store_nt(rs_h, svoff, so, hn);
                }
            }
            // GI: the next step's tile, requested BEHIND the cell update (more live registers across it did not fit): it lands
            // under the x-free barrier phase and the next step's MFMAs
            if (GI && t + 1 < a.T) gfetch(E0);
        };
        if (HALVES) {
            if (part == 0) cell(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            else cell(std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});
        } else if (part == 0) {
            // (PARTS == 4, H = 64: the round-4 form, textually: routed through the lambda above hipcc spills 12-20 registers here)
            if (GI) {      // the gi tile first: its 48 registers are free before the 192 partial sums arrive
#pragma unroll
                for (int g = 0; g < 3; g++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[0][g][e] += gn[g][e];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < PARTS - 1; p++) {
                const float *src = xch + (size_t)p * 64 * 64;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[0][g][e] += src[(g * 16 + e) * 64 + lane];
            }
            rsrc_t rs_r, rs_z, rs_n, rs_g, rs_h;
            if (SAVE) {
                const size_t so_t = (size_t)t * B * H;
                const uint32_t sbytes = (uint32_t)a.B * (uint32_t)H * 4u;
                rs_r = make_rsrc(a.sv_r + so_t, sbytes); rs_z = make_rsrc(a.sv_z + so_t, sbytes); rs_n = make_rsrc(a.sv_n + so_t, sbytes);
                rs_g = make_rsrc(a.sv_g + so_t, sbytes); rs_h = make_rsrc(a.sv_h + so_t, sbytes);
            }
            CellPair cp;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int hidx = row * HS + chunk * 32 + li;
                if ((e & 1) == 0)
                    cp = gru_cell_pair((osk::f2){acc[0][0][e], acc[0][0][e + 1]}, (osk::f2){acc[0][1][e], acc[0][1][e + 1]},
                                       (osk::f2){acc[0][2][e], acc[0][2][e + 1]}, (osk::f2){acc[0][3][e], acc[0][3][e + 1]},
                                       (osk::f2){hl[hidx], hl[hidx + HS]}, nb_r, nb_z, nb_n, b_hn);
                const float r = cp.r[e & 1], z = cp.z[e & 1], n = cp.n[e & 1], ghn = cp.ghn[e & 1], hn = cp.hn[e & 1];
                hn_buf[hidx] = hn;
                if (SAVE) {
                    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)((e & 3) + 8 * (e >> 2)) * (uint32_t)H * 4u);
                    osk::buf_store_nt(rs_r, svoff, so, r); osk::buf_store_nt(rs_z, svoff, so, z); osk::buf_store_nt(rs_n, svoff, so, n);
                    osk::buf_store_nt(rs_g, svoff, so, ghn); osk::buf_store_nt(rs_h, svoff, so, hn);
                }
            }
        }
        OSL_TS(4)                                            // part 0: sum of the partials, cell update, saved activations
        if (!GI && t + 1 < a.T) xstage(xl2 + ((t + 1) & 1) * BM * XS);
        lds_barrier();   // h_t and x_{t+1} complete (and the exchange buffers free again)
        OSL_TS(5)                                            // x tile staged, barrier
        if (STACK && a.seq_out) publish(hn_buf, t);          // the next layer's workgroup of this tile may take step t now
    }
#ifdef OS_LAYER_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("gru_layer_split_kernel<%d> K=%d T=%d cycles per step (wave 0 = chunk 0, part 0): write-back %llu | GEMM slice %llu | exchange + barrier %llu | cell update %llu | x stage + barrier %llu | sum %llu; prologue %llu\n",
               NCH, a.K, a.T, ts_sum[1] / a.T, ts_sum[2] / a.T, ts_sum[3] / a.T, ts_sum[4] / a.T, ts_sum[5] / a.T,
               (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5]) / a.T, ts_pro);
#endif
    const float *hT = hl2 + (a.T & 1) * BM * HS;
    if (!STACK && a.seq_out) write_back(hT, a.seq_out + (size_t)(a.T - 1) * H * B);
    if (a.h_last) write_back(hT, a.h_last);
}

template <int NCH, bool SAVE, bool GI = false>
__global__ __launch_bounds__(512, 1) void gru_layer_split_kernel(const LayerArgs a)
{
    split_layer_body<NCH, false, SAVE, GI>(a, StackSync{nullptr, nullptr});
}

// gi [N][3H] = rows [N][K] . W_ih^T for os_gru_forward_windows: the input half of layer 0's gate GEMM once per ROW of the stream
// (the windowed form recomputes it for each of the T windows that contain the row: 144,384 of 832,512 flop per GRU step at
// RNN(188,128,4), gru/gru_test.py:138-140).  Workgroup = 64 rows x all 3H columns: wave c = the 32 hidden units of chunk c for
// the three gates, two 32-row blocks; the row tile staged once in LDS (coalesced), A fragments by ds_read (odd pitch), the packed
// x-part fragments of the layer-0 image as in the layer kernels.  No bias here: the cell update folds the biases in.
struct GiArgs {
    int N, K, H, KPx, KPh;
    const float *rows;         // [N][K] row-major (the caller's row stream)
    const float *w;            // packed layer-0 weights (chunk-major, x part first)
    float *gi;                 // [N][3H]
};
// RBW = 32-row blocks per workgroup: 2 when that still gives every CU two workgroups, else 1 (8,201 rows: 257 workgroups, not 129)
template <int RBW>
__global__ __launch_bounds__(256, 2) void gru_gi_kernel(const GiArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [32 RBW][XS]
    constexpr int BM = 32 * RBW;
    const int lane = threadIdx.x & 63, chunk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NCH = a.H / 32;
    const int li = lane & 31, lh = lane >> 5;
    const int XS = 2 * a.KPx + 1, row0 = blockIdx.x * BM;
    // stage: wave w takes rows w, w + 4, ...; its 64 lanes walk a row (K floats apart: consecutive lanes, consecutive addresses)
    for (int r = chunk; r < BM; r += 4) {
        const int g = row0 + r;
        const float *src = a.rows + (size_t)(g < a.N ? g : a.N - 1) * a.K;
        for (int k = lane; k < XS; k += 64) smem[r * XS + k] = (k < a.K && g < a.N) ? src[k] : 0.f;
    }
    __syncthreads();
    for (int c = chunk; c < NCH; c += 4) {
        f32x16 acc[RBW][4];
#pragma unroll
        for (int rb = 0; rb < RBW; rb++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rb][g][e] = 0.f;
        const float *wx = a.w + (size_t)c * chunk_floats(a.KPx, a.KPh);
        mfma_part<RBW, true>(acc, wx, a.KPx, lane, [&](int q, int rb) { return smem[(rb * 32 + li) * XS + 2 * q + lh]; });
        const rsrc_t ro = make_rsrc(a.gi, (uint32_t)a.N * 3u * (uint32_t)a.H * 4u);       // rows past N fall outside: dropped
#pragma unroll
        for (int rb = 0; rb < RBW; rb++) {
            const uint32_t vo = (uint32_t)(((size_t)(row0 + rb * 32 + 4 * lh) * 3 * a.H + c * 32 + li) * 4);
#pragma unroll
            for (int g = 0; g < 3; g++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    osk::buf_store(ro, vo, (uint32_t)(((e & 3) + 8 * (e >> 2)) * 3 * a.H + g * a.H) * 4u, acc[rb][g][e]);
        }
    }
}

// Small batches, several layers: ONE launch, blockIdx.y = layer, the layers of a tile pipelined against each other (round 4).
// At B <= 32 x (CUs / layers) every (layer, tile) workgroup is resident at once, so layer l can run step t while layer l - 1 runs
// step t + 2: a stack of L layers takes about T + 2 (L - 1) step times instead of L T (VERDICT r3 next 3 proposed this
// layer-diagonal order for the 8,192-window training batch, where every CU already holds a tile and nothing is idle: section
// 4.4; it pays where CUs ARE idle -- config 5's 128 trajectories, the reference's own batches of 64 and 1).  Workgroups are
// dispatched in linear order (layer-major), a consumer only ever waits for a workgroup with a smaller id, and every wait is
// bounded, so a producer that never shows up costs wrong numbers, not a hung device.
struct StackArgs {
    int n, tiles;
    uint32_t *flags;             // [n][tiles] progress counters, zeroed before the launch
    int32_t *err, *err_local;    // the context's error word (host-mapped) and its device twin
    uint32_t max_polls;
    int drop_layer, drop_step;   // tests: layer drop_layer stops publishing from step drop_step on (-1: never)
    LayerArgs layer[8];
};
template <int NCH, bool SAVE>
__global__ __launch_bounds__(512, 1) void gru_stack_kernel(const StackArgs sa)
{
    const int l = blockIdx.y;
    uint32_t *mine = sa.flags + (size_t)l * sa.tiles + blockIdx.x;
    split_layer_body<NCH, true, SAVE>(sa.layer[l], StackSync{l > 0 ? mine - sa.tiles : nullptr, mine, sa.err, sa.err_local, sa.max_polls,
                                                             l == sa.drop_layer ? sa.drop_step : -1});
}

// H = 128 small-batch variant with the INPUT half of the gate GEMM running ahead of the recurrence.
// gi = x_t W_ih^T does not depend on h, so nothing forces it into the same barrier-delimited phase as the recurrent half:
// waves 4-7 ("x waves", one per 32-column chunk) compute it one to two steps AHEAD and hand the three partial gates to
// waves 0-3 ("h waves") through LDS; the h waves do the recurrent half and the cell update.  What that buys: in
// gru_layer_split_kernel a step was [both halves' MFMAs] -> barrier -> [cell update on four waves while the other four
// sat at the barrier] (in-kernel timestamps at the training batch: 28 k cycles + 5.6 k of a 36.5 k-cycle step, for 24.6 k
// cycles of MFMAs); here the x waves' MFMAs fill the SIMDs while the h waves are in the cell update, and the recurrent
// MFMAs of step 0 (h_{-1} = 0) are skipped outright.  A step is three barrier-delimited phases:
//   1   h waves: recurrent MFMAs of step t                    x waves: rest of the input MFMAs of step t+1, h_{t-1} -> seq_out
//   W   h waves: pick up the partial gates P(t), stage x_{t+2} into the x tile (its registers were loaded in phase 2 of t-1)
//   2   h waves: request x_{t+3}, cell update of step t       x waves: publish P(t+1), first part of the input MFMAs of t+2
// LDS: h double buffer [2][32][129] | x tile [32][2 KPx + 1] | partial gates [4][48][64]  (96 KB at K = 128).
__global__ __launch_bounds__(512, 1) void gru_layer_ahead_kernel(const LayerArgs a)
{
    constexpr int H = 128, HS = H + 1, BM = 32, XE = 24;          // x staging: 8 inputs per pass and thread row, K <= 192
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_x = wave >= 4;
    const int chunk = wave & 3;
    const int tile_row0 = blockIdx.x * BM;
    const int li = lane & 31, lh = lane >> 5;
    const size_t B = (size_t)a.B;
    const int KPx = a.KPx, XS = 2 * KPx + 1;
    float *hl2 = smem, *xl = smem + 2 * BM * HS, *xch = xl + BM * XS + (size_t)chunk * 48 * 64;

    for (int i = threadIdx.x; i < BM * HS; i += 512) hl2[i] = 0.f;   // h0 = 0 (gru/gru_model.py:27)
    for (int i = threadIdx.x; i < BM * XS; i += 512) xl[i] = 0.f;    // the pad column of an odd input width stays zero

    const float *wx = a.w + (size_t)chunk * chunk_floats(a.KPx, a.KPh);
    const float *wh = wx + (size_t)a.KPx * 3 * 64;
    const float *bias = wh + (size_t)a.KPh * 3 * 64;
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];
    const uint32_t rowB = (uint32_t)a.B * 4u;
    const int g0 = tile_row0 + li;
    const int growc = g0 < a.B ? g0 : a.B - 1;
    // x tile staging by the h waves: thread -> (input k0 + 8 e, row li): 128-byte segments of the [K][B] stream; inputs past K
    // read zero through the descriptor's range check and are not written
    // With xs_btf the layer reads the caller's (B, T, K) tensor itself (no [T][K][B] copy of the input: 27 us and 123 MB per
    // training step at 8192 x 10 x 188): thread -> (row (tid >> 3) & 31, inputs (tid & 7) + 8 e), 32-byte runs of a window's row.
    const int xk0 = (threadIdx.x >> 5) & 7;
    const uint32_t xsoff = (uint32_t)growc * 4u + (uint32_t)xk0 * rowB;
    const int brow = (threadIdx.x >> 3) & 31, bk0 = threadIdx.x & 7;
    const int bg = tile_row0 + brow < a.B ? tile_row0 + brow : a.B - 1;
    const uint32_t xboff = (uint32_t)(((size_t)bg * a.T * a.K + bk0) * 4);
    float xr[XE];
    auto xfetch = [&](int s) {
        if (s >= a.T) return;
        if (a.xs_btf) {
            const rsrc_t rx = make_rsrc(a.xs, (uint32_t)((size_t)a.B * a.T * a.K * 4));       // host: < 4 GiB
            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)s * (uint32_t)a.K * 4u);
#pragma unroll
            for (int e = 0; e < XE; e++) {
                if (e * 8 >= a.K) break;
                xr[e] = bk0 + e * 8 < a.K ? buf_load(rx, xboff + (uint32_t)e * 32u, so) : 0.f;
            }
            return;
        }
        const rsrc_t rx = make_rsrc(a.xs + (size_t)s * a.K * B, (uint32_t)a.K * rowB);
#pragma unroll
        for (int e = 0; e < XE; e++) {
            if (e * 8 >= a.K) break;
            xr[e] = buf_load(rx, xsoff, __builtin_amdgcn_readfirstlane((uint32_t)(e * 8) * rowB));
        }
    };
    auto xstage = [&]() {
        if (a.xs_btf) {
#pragma unroll
            for (int e = 0; e < XE; e++) {
                if (e * 8 >= a.K) break;
                if (bk0 + e * 8 < a.K) xl[brow * XS + bk0 + e * 8] = xr[e];
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < XE; e++) {
            if (e * 8 >= a.K) break;
            if (xk0 + e * 8 < a.K) xl[li * XS + xk0 + e * 8] = xr[e];
        }
    };
    const int SP = KPx < 24 ? KPx : 24;                          // input k-pairs done in phase 2 (beside the cell update)
    f32x16 acc[1][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[0][g][e] = 0.f;
    };
    auto xpart = [&](int lo, int hi) {
        if (hi > lo)
            mfma_part<1, true, 8>(acc, wx + (size_t)lo * 3 * 64, hi - lo, lane, [&](int q, int) { return xl[li * XS + 2 * (q + lo) + lh]; });
    };
    auto publish = [&]() {                                       // r, z, n_x partial gates -> LDS
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int e = 0; e < 16; e++) xch[(g * 16 + e) * 64 + lane] = acc[0][g][e];
    };
    auto write_back = [&](const float *hsrc, float *dst, int first, int stride) {
        for (int i = first; i < BM * H; i += stride) {
            const int row = i % BM, k = i / BM;
            const int g = tile_row0 + row;
            if (g < a.B) dst[(size_t)k * B + g] = hsrc[row * HS + k];
        }
    };

    // ---- prologue: P(0) published, first part of the input MFMAs of step 1 in the x waves' accumulators, x_1 in the tile,
    // x_2 in the h waves' registers ----
    __syncthreads();
    if (!is_x) { xfetch(0); xstage(); }
    __syncthreads();
    if (is_x) { zero_acc(); xpart(0, KPx); publish(); }
    else xfetch(1);
    lds_barrier();
    if (!is_x && 1 < a.T) xstage();
    lds_barrier();
    if (is_x) { zero_acc(); if (1 < a.T) xpart(0, SP); }
    else xfetch(2);

    for (int t = 0; t < a.T; t++) {
        const float *hl = hl2 + (t & 1) * BM * HS;          // h_{t-1}
        float *hn_buf = hl2 + ((t + 1) & 1) * BM * HS;       // h_t
        // ---- phase 1 ----
        if (is_x) {
            if (t > 0 && a.seq_out) {
                // h_{t-1} -> seq_out [T][H][B]: sixteen LDS reads in flight, then sixteen 128-byte-segment stores (thread ->
                // hidden unit xk0 + 8 e, row li), ahead of the MFMAs so that they drain underneath them
                const rsrc_t rq = make_rsrc(a.seq_out + (size_t)(t - 1) * H * B, (uint32_t)H * rowB);
                float hv[16];
#pragma unroll
                for (int e = 0; e < 16; e++) hv[e] = hl[li * HS + xk0 + 8 * e];
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (g0 < a.B) osk::buf_store(rq, xsoff, (uint32_t)(8 * e) * rowB, hv[e]);
            }
            if (t + 1 < a.T) xpart(SP, KPx);
        } else {
            zero_acc();
            if (t > 0) mfma_part<1, false, 8>(acc, wh, a.KPh, lane, [&](int q, int) { return hl[li * HS + 2 * q + lh]; });
        }
        lds_barrier();   // every wave is done with the x tile; P(t) was published before the previous barrier
        // ---- window ----
        if (!is_x) {
#pragma unroll
            for (int g = 0; g < 3; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[0][g][e] += xch[(g * 16 + e) * 64 + lane];
            if (t + 2 < a.T) xstage();
        }
        lds_barrier();   // P(t) consumed, x_{t+2} staged
        // ---- phase 2 ----
        if (is_x) {
            if (t + 1 < a.T) publish();
            zero_acc();
            if (t + 2 < a.T) xpart(0, SP);
        } else {
            xfetch(t + 3);
            const uint32_t svbytes = (uint32_t)((size_t)a.T * B * H * 4);       // host: < 4 GiB
            const rsrc_t rs_r = make_rsrc(a.sv_r, svbytes), rs_z = make_rsrc(a.sv_z, svbytes), rs_n = make_rsrc(a.sv_n, svbytes),
                         rs_g = make_rsrc(a.sv_g, svbytes), rs_h = make_rsrc(a.sv_h, svbytes);
            const uint32_t svoff = (uint32_t)(((size_t)(tile_row0 + 4 * lh) * H + chunk * 32 + li) * 4);
            CellPair cp;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int hidx = row * HS + chunk * 32 + li;
                if ((e & 1) == 0)                                    // element pairs (gru_cell_pair): e + 1 is the next row
                    cp = gru_cell_pair((osk::f2){acc[0][0][e], acc[0][0][e + 1]}, (osk::f2){acc[0][1][e], acc[0][1][e + 1]},
                                       (osk::f2){acc[0][2][e], acc[0][2][e + 1]}, (osk::f2){acc[0][3][e], acc[0][3][e + 1]},
                                       (osk::f2){hl[hidx], hl[hidx + HS]}, nb_r, nb_z, nb_n, b_hn);
                const float r = cp.r[e & 1], z = cp.z[e & 1], n = cp.n[e & 1], ghn = cp.ghn[e & 1], hn = cp.hn[e & 1];
                hn_buf[hidx] = hn;
                if (a.sv_r && tile_row0 + row < a.B) {
                    // one per-lane offset register for the 80 stores of a step (flat addresses: two registers per store in flight)
                    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(((size_t)t * B + (e & 3) + 8 * (e >> 2)) * H * 4));
                    osk::buf_store_nt(rs_r, svoff, so, r); osk::buf_store_nt(rs_z, svoff, so, z); osk::buf_store_nt(rs_n, svoff, so, n);
                    osk::buf_store_nt(rs_g, svoff, so, ghn); osk::buf_store_nt(rs_h, svoff, so, hn);
                }
            }
        }
        lds_barrier();   // h_t complete
    }
    const float *hT = hl2 + (a.T & 1) * BM * HS;
    if (a.seq_out) write_back(hT, a.seq_out + (size_t)(a.T - 1) * H * B, threadIdx.x, 512);
    if (a.h_last) write_back(hT, a.h_last, threadIdx.x, 512);
}

// Re-pack the torch-layout weights (W_ih [3H][K], W_hh [3H][H], b_ih [3H], b_hh [3H]) into fragment order;
// all layers in one launch (blockIdx.z = layer): the training step re-packs every optimisation step.
// blockIdx.z in [n, 2n) (launched only when vdst is set): the same layers' image for gru_vec_kernel (layout in VecArgs) -- both
// images come from ONE snapshot of the caller's flat weights, taken by ONE launch of os_gru_load.
__host__ __device__ inline size_t vec_layer_floats(int K, int H) { return (size_t)((K + 15) & ~15) * 3 * H + (size_t)H * 3 * H + 3 * H + H; }
struct PackAll {
    int n, H;
    int K[16];
    const float *Wih[16], *Whh[16], *bih[16], *bhh[16];
    float *dst[16];
    float *vdst[16];
};
__global__ void gru_pack_all_kernel(const PackAll a)
{
    if ((int)blockIdx.z >= a.n) {
        const int l = blockIdx.z - a.n, H = a.H, R = 3 * H, K = a.K[l];
        if (l >= a.n) return;
        float *d = a.vdst[l];
        const float *Wih = a.Wih[l], *Whh = a.Whh[l], *bih = a.bih[l], *bhh = a.bhh[l];
        const int nih = ((K + 15) & ~15) * R, nhh = H * R;
        const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
        for (int i = blk * blockDim.x + threadIdx.x; i < nih + nhh + R + H; i += nblk * blockDim.x) {
            float v;
            if (i < nih) { const int k = (i / (16 * R)) * 16 + 4 * (i & 3) + ((i >> 2) & 3), r = (i >> 4) % R; v = k < K ? Wih[(size_t)r * K + k] : 0.f; }
            else if (i < nih + nhh) {     // [quad i4][gate g][unit u][column quarter c][4]: W_hh[g H + u][c H/4 + 4 i4 + e]
                const int j = i - nih, e = j & 3, c = (j >> 2) & 3, u = (j >> 4) % H, g = (j / (16 * H)) % 3, i4 = j / (48 * H);
                v = Whh[(size_t)(g * H + u) * H + c * (H / 4) + 4 * i4 + e];
            }
            else if (i < nih + nhh + R) { const int r = i - nih - nhh; v = bih[r] + (r < 2 * H ? bhh[r] : 0.f); }
            else v = bhh[2 * H + (i - nih - nhh - R)];
            d[i] = v;
        }
        return;
    }
    const int l = blockIdx.z;
    const int K = a.K[l], H = a.H, KPx = (K + 1) / 2, KPh = H / 2, chunk = blockIdx.x;
    const float *Wih = a.Wih[l], *Whh = a.Whh[l], *bih = a.bih[l], *bhh = a.bhh[l];
    float *d = a.dst[l] + (size_t)chunk * chunk_floats(KPx, KPh);
    const int nx = KPx * 3 * 64, nh = KPh * 3 * 64;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nx + nh + 128; i += gridDim.y * blockDim.x) {
        float v;
        if (i < nx + nh) {
            const bool isx = i < nx;
            const int j = isx ? i : i - nx;
            const int lane = j & 63, g = (j >> 6) % 3, kp = (j >> 6) / 3;
            const int k = 2 * kp + (lane >> 5), col = g * H + chunk * 32 + (lane & 31);
            if (isx) v = (k < K) ? Wih[(size_t)col * K + k] : 0.f;
            else v = Whh[(size_t)col * H + k];
        } else {
            const int j = i - nx - nh, which = j >> 5, c = chunk * 32 + (j & 31);
            if (which == 0) v = bih[c] + bhh[c];
            else if (which == 1) v = bih[H + c] + bhh[H + c];
            else if (which == 2) v = bih[2 * H + c];
            else v = bhh[2 * H + c];
        }
        d[i] = v;
    }
}

// fc + sigmoid on the last hidden state (gru/gru_model.py:43-48).  h_last [H][B] -> out [B][C].
// blockIdx.y = group of eight classes.  A 256-thread workgroup covers 64 trajectories: wave q walks a quarter of the hidden
// units (the loads of a quarter are all in flight together: one memory round trip instead of a chain of them), the four
// partial sums meet in LDS.  The eight rows of fc weights are staged with coalesced reads, transposed to [H][8], so that
// the eight weights of a k are two wave-uniform ds_read_b128.
__global__ __launch_bounds__(256) void gru_head_kernel(int B, int H, int C, const float *h_last, const float *fcw, const float *fcb,
                                int use_sigmoid, float *out)
{
    extern __shared__ __attribute__((aligned(16))) float sw[];     // [H][8] weights | [8] bias | [3][8][64] partial sums
    const int c0 = blockIdx.y * 8;
    for (int i = threadIdx.x; i < 8 * H; i += blockDim.x) {
        const int j = i / H, k = i % H;                              // consecutive threads: consecutive k of one class row
        sw[k * 8 + j] = c0 + j < C ? fcw[(size_t)(c0 + j) * H + k] : 0.f;
    }
    if (threadIdx.x < 8) sw[8 * H + threadIdx.x] = c0 + threadIdx.x < C ? fcb[c0 + threadIdx.x] : 0.f;
    __syncthreads();
    float *ps = sw + 8 * H + 8;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane, bc = b < B ? b : B - 1;
    const int k0 = q * (H / 4), k1 = q == 3 ? H : k0 + H / 4;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = 0.f;
#pragma unroll 16
    for (int k = k0; k < k1; k++) {
        const float hv = h_last[(size_t)k * B + bc];
        const float4 w0 = *reinterpret_cast<const float4 *>(&sw[k * 8]), w1 = *reinterpret_cast<const float4 *>(&sw[k * 8 + 4]);
        s[0] = fmaf(w0.x, hv, s[0]); s[1] = fmaf(w0.y, hv, s[1]); s[2] = fmaf(w0.z, hv, s[2]); s[3] = fmaf(w0.w, hv, s[3]);
        s[4] = fmaf(w1.x, hv, s[4]); s[5] = fmaf(w1.y, hv, s[5]); s[6] = fmaf(w1.z, hv, s[6]); s[7] = fmaf(w1.w, hv, s[7]);
    }
    if (q > 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) ps[((q - 1) * 8 + j) * 64 + lane] = s[j];
    }
    __syncthreads();
    if (q == 0 && b < B) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = ((s[j] + ps[j * 64 + lane]) + (ps[(8 + j) * 64 + lane] + ps[(16 + j) * 64 + lane])) + sw[8 * H + j];
            if (c0 + j < C) out[(size_t)b * C + c0 + j] = use_sigmoid ? sigmoidf_(v) : v;
        }
    }
}


// ---- B <= 4: the whole model in ONE single-workgroup launch on the vector pipe (round 4) --------------------------------------
// The reference's own evaluation loop calls the model on ONE window at a time and reports the time of that call
// (gru/gru_test.py:157, :171-177: DataLoader(batch_size=1), `computation_time`).  A 32-row MFMA tile is 31/32 padding there and a
// step costs the full 12.6 us of tile MFMAs; as matrix-vector products the same step is 49 k multiply-adds.  One workgroup of 512
// threads (eight waves, 256 registers each) runs layer after layer:
//   phase A  gi[n][r] = b + W_ih[r][:] x_n for ALL N = B T columns at once (x does not depend on this layer's h) on
//            v_mfma_f32_16x16x4_f32: wave = three (H = 64: two) blocks of 16 gate rows x up to three blocks of 16 columns over the whole
//            K; each weight is read once (16-byte loads of a fragment-ordered image), the x fragments are ds_read_b32.  (First
//            version on the vector pipe with the x broadcast inside v_fmac_f32_dpp: 22 k cycles per layer against 7-9 k, DPP
//            multiply-adds issue at ~5 cycles per wave whatever the occupancy);
//   phase B  per step: thread = (hidden unit u, column quarter c) of 4 H threads; its 3 gates x H / 4 recurrent weights stay in
//            REGISTER PAIRS for the whole layer (H = 128: 96 VGPRs, 196 KB of the CU's 512 KB register file), h_{t-1} arrives as
//            16-byte LDS reads, the products are v_pk_fma_f32 on six independent sums; the four quarters of a unit sit in one DPP
//            quad, so two quad_perm adds finish the three gate sums and the same lanes do the cell update: ONE barrier per step
//            (first version: thread = (gate row, column half), partial sums through LDS, a 128-thread cell update between two
//            barriers: 1.6 k cycles per step);
// No cross-workgroup traffic; the head (fc + sigmoid) and h_T of every layer in torch layout come out of the same launch.
struct VecArgs {
    int B, T, K0, L, C, use_sigmoid;
    const float *x;              // (B, T, K0) as the caller passes it
    const float *wvec;           // per layer: W_ih as [ceil(K / 16)][3H][4 kk][4 i], k = 16 Q + 4 i + kk (zero padded) | W_hh as [H / 16][3 gates][H units][4 column quarters][4] | b_gi [3H] (b_ih + b_hh for r, z; b_in) | b_hn [H]
    const float *fcw, *fcb;
    float *out;                  // [B][C]
    float *h_last;               // [L][B][H] or null
};
constexpr int VEC_NMAX = 48, VEC_BMAX = 4, VEC_THREADS = 512;     // eight waves, two per SIMD: 256 registers each (twelve waves: 168, spills)
__device__ __forceinline__ float4 buf_load4(rsrc_t r, uint32_t voff, uint32_t soff)
{
    // (cast the whole vector: __builtin_bit_cast of ONE element of a vector-typed value reads element 0 whatever the index, hipcc 7.0)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
}
// Phase A of gru_vec_kernel: gi[n][m] = b[m] + sum_k W_ih[m][k] x[k][n] for NB blocks of 16 columns on v_mfma_f32_16x16x4_f32; wave = MB
// blocks of 16 gate rows over the whole K.  KQ > 0 (the reference's widths: K = 16 KQ after padding): ALL of the wave's weight
// fragments are requested up front (KQ MB 16-byte loads, at most 96 registers) and the loop over the K groups is unrolled, so the
// wait counts are exact and the first MFMA starts when the first fragment lands.  (A two-ahead prefetch that rotated its registers
// made hipcc wait for the youngest load at every copy.)  KQ = 0: any K, one group ahead in two alternating register sets.
template <int H, int NB, int KQ, class Mid>
__device__ __forceinline__ void vec_phase_a(const float *xin, float *gi, const float *wih, const float *bgi, int K, int N, int NP, int tid, Mid &&mid)
{
    constexpr int R = 3 * H, NWV = VEC_THREADS / 64, MB = (R / 16 + NWV - 1) / NWV;   // blocks of 16 gate rows per wave (H = 128: 3 on eight waves; 64: 2 on six)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, kk = lane >> 4;
    const int K16 = (K + 15) >> 4;
    if (wave * MB >= R / 16) {                                       // (H = 64: waves 6, 7 have no rows; their phase-B registers are still requested)
        mid();
        return;
    }
    const rsrc_t ri = make_rsrc(wih, (uint32_t)K16 * R * 64u);
    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments: lane (gate row m0 + l16, kk) takes the four inputs k = 16 Q + 4 i + kk of its row in one 16-byte load
    uint32_t wvo[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) wvo[mb] = (uint32_t)(((wave * MB + mb) * 16 + l16) * 4 + kk) * 16u;
    // A fragments: lane (column nb 16 + l16, kk) reads x[k][n]; padded inputs carry weight 0 and take any finite x
    auto load_x = [&](int Q, float (&xa)[NB][4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int k = 16 * Q + 4 * i + kk;
            k = k < K ? k : K - 1;
#pragma unroll
            for (int nb = 0; nb < NB; nb++) xa[nb][i] = xin[k * NP + nb * 16 + l16];
        }
    };
    auto mfmas = [&](const float4 (&wc)[MB], const float (&xa)[NB][4]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) {
                const float wv = i == 0 ? wc[mb].x : i == 1 ? wc[mb].y : i == 2 ? wc[mb].z : wc[mb].w;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[nb][i], wv, acc[mb][nb], 0, 0, 0);
            }
    };
    if constexpr (KQ > 0) {
        float4 wall[KQ][MB];
#pragma unroll
        for (int Q = 0; Q < KQ; Q++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) wall[Q][mb] = buf_load4(ri, wvo[mb], (uint32_t)(Q * R) * 64u);
        __builtin_amdgcn_sched_barrier(0);                           // request order = wait order
        float xa[2][NB][4];
        load_x(0, xa[0]);
#pragma unroll
        for (int Q = 0; Q < KQ; Q++) {
            if (Q + 1 < KQ) load_x(Q + 1, xa[(Q + 1) & 1]);
            // the caller's requests for phase B (the recurrent weights) go out once enough fragment registers are free again: they land
            // underneath the remaining MFMAs
            if (Q == KQ - (KQ >= 8 ? 3 : 2)) {
                __builtin_amdgcn_sched_barrier(0);                   // not earlier: the fragment registers it reuses are still live
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfmas(wall[Q], xa[Q & 1]);
        }
    } else {
        float4 wA[MB], wB[MB];
        float xA[NB][4], xB[NB][4];
#pragma unroll
        for (int mb = 0; mb < MB; mb++) wA[mb] = buf_load4(ri, wvo[mb], 0);
        load_x(0, xA);
        for (int Q = 0; Q < K16; Q += 2) {                           // (a group past the image reads zero weights)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) wB[mb] = buf_load4(ri, wvo[mb], __builtin_amdgcn_readfirstlane((uint32_t)((Q + 1) * R) * 64u));
            load_x(Q + 1, xB);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(wA, xA);
#pragma unroll
            for (int mb = 0; mb < MB; mb++) wA[mb] = buf_load4(ri, wvo[mb], __builtin_amdgcn_readfirstlane((uint32_t)((Q + 2) * R) * 64u));
            load_x(Q + 2, xA);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(wB, xB);
        }
        mid();
    }
    // D: lane holds gi[n = nb 16 + 4 kk + e][m = m0 + l16]
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const int m = (wave * MB + mb) * 16 + l16;
        const float bm = bgi[m];
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int n = nb * 16 + 4 * kk + e;
                if (n < N) gi[n * R + m] = acc[mb][nb][e] + bm;
            }
    }
}
template <int H, int NB, class Mid>
__device__ __forceinline__ void vec_phase_a_k(const float *xin, float *gi, const float *wih, const float *bgi, int K, int N, int NP, int tid, Mid &&mid)
{
    const int K16 = (K + 15) >> 4;
    if (K16 == 12 && NB == 1) vec_phase_a<H, NB, (NB == 1 ? 12 : 0)>(xin, gi, wih, bgi, K, N, NP, tid, mid);   // (144 weight registers at H = 128: one column block only)
    else if (K16 == 8) vec_phase_a<H, NB, 8>(xin, gi, wih, bgi, K, N, NP, tid, mid);
    else if (K16 == 4) vec_phase_a<H, NB, 4>(xin, gi, wih, bgi, K, N, NP, tid, mid);
    else vec_phase_a<H, NB, 0>(xin, gi, wih, bgi, K, N, NP, tid, mid);
}

template <int H>
__global__ __launch_bounds__(VEC_THREADS) void gru_vec_kernel(const VecArgs a)
{
    constexpr int R = 3 * H, HQ = H / 4, NTB = 4 * H;                  // phase B: NTB threads = (unit, column quarter), HQ columns each
    static_assert(NTB <= VEC_THREADS && HQ % 4 == 0 && R % 192 == 0, "H = 64 or 128");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int B = a.B, T = a.T, N = B * T, NP = (N + 3) & ~3;
    const int KA = a.K0 > H ? a.K0 : H;
    float *xbuf0 = smem, *xbuf1 = smem + KA * NP;                      // layer input [k][NP] (column n = b T + t), ping-pong
    float *gi = xbuf1 + H * NP;                                        // [NP][R]
    float *hbuf = gi + NP * R;                                         // [2][B][H] h double buffer
    const int tid = threadIdx.x;
    const int pu = tid >> 2, pc = tid & 3;                            // phase B: hidden unit, column quarter (lanes 4 u .. 4 u + 3 = one DPP quad)
    const bool pb_active = tid < NTB;                                  // wave-uniform

#ifdef OS_LAYER_TS
    unsigned long long vts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vprev = __builtin_readcyclecounter();
#define VTS(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now = __builtin_readcyclecounter(); vts[i] += now - vprev; vprev = now; __builtin_amdgcn_sched_barrier(0); }
#else
#define VTS(i)
#endif
    for (int i = tid; i < KA * NP; i += VEC_THREADS) xbuf0[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < N * a.K0; i += VEC_THREADS) xbuf0[(i % a.K0) * NP + i / a.K0] = a.x[i];
    __syncthreads();
    VTS(0)                                                             // x -> LDS

    const float *w = a.wvec;
    for (int l = 0; l < a.L; l++) {
        const int K = l == 0 ? a.K0 : H;
        const float *wih = w, *whh = w + (size_t)((K + 15) & ~15) * R, *bgi = whh + (size_t)H * R, *bhn = bgi + R;
        const float *xin = (l & 1) ? xbuf1 : xbuf0;
        float *xout = (l & 1) ? xbuf0 : xbuf1;
        const bool last = l == a.L - 1;
        // this thread's recurrent weights -> register pairs for the whole layer: 16-byte loads, 1 KB contiguous per wave and load, one
        // per-lane offset with the (quad, gate) in the scalar offset (flat addresses: hipcc kept a 64-bit address pair per weight and
        // spilled them).  Requested from inside phase A (its `mid` hook): before it, the extra live registers spill; after it, the
        // round trip is exposed
        osk::f2 wr[3][HQ / 2];
        auto load_wr = [&]() {           // (every thread, also the idle ones of phase B: a conditional definition keeps the array live --
                                         // and spilled -- across all of phase A; their offsets stay inside the image or read zero)
            const rsrc_t rw = make_rsrc(whh, (uint32_t)H * R * 4u);
#pragma unroll
            for (int i4 = 0; i4 < HQ / 4; i4++)
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    const float4 v = buf_load4(rw, (uint32_t)tid * 16u, (uint32_t)((i4 * 3 + g) * H) * 64u);
                    wr[g][2 * i4] = (osk::f2){v.x, v.y}; wr[g][2 * i4 + 1] = (osk::f2){v.z, v.w};
                }
        };
        // ---- phase A ----
        {
            const int NB = (NP + 15) >> 4;
            if (NB == 1) vec_phase_a_k<H, 1>(xin, gi, wih, bgi, K, N, NP, tid, load_wr);
            else if (NB == 2) vec_phase_a_k<H, 2>(xin, gi, wih, bgi, K, N, NP, tid, load_wr);
            else vec_phase_a_k<H, 3>(xin, gi, wih, bgi, K, N, NP, tid, load_wr);
        }
        VTS(1)                                                         // phase A
        VTS(2)
        __syncthreads();                                               // gi complete
        VTS(3)                                                         // barrier behind phase A
        // ---- phase B ----
        float hprev[VEC_BMAX] = {0.f, 0.f, 0.f, 0.f};                  // h_{t-1}[b][pu]; h0 = 0 (gru/gru_model.py:27)
        const float my_bhn = pb_active ? bhn[pu] : 0.f;
        for (int t = 0; t < T; t++) {
            const float *hc = hbuf + (t & 1) * VEC_BMAX * H;           // h_{t-1}
            float *hn_buf = hbuf + ((t + 1) & 1) * VEC_BMAX * H;       // h_t
            if (pb_active) {
#pragma unroll
                for (int b = 0; b < VEC_BMAX; b++) {
                    if (b < B) {
                        const int n = b * T + t;
                        const float gir = gi[n * R + pu], giz = gi[n * R + H + pu], gin = gi[n * R + 2 * H + pu];   // in flight under the products
                        float sr = 0.f, sz = 0.f, sn = 0.f;
                        if (t > 0) {
                            typedef float f4 __attribute__((ext_vector_type(4)));
                            osk::f2 ar[2] = {{0.f, 0.f}, {0.f, 0.f}}, az[2] = {{0.f, 0.f}, {0.f, 0.f}}, an[2] = {{0.f, 0.f}, {0.f, 0.f}};
                            const f4 *hb = reinterpret_cast<const f4 *>(hc + b * H + pc * HQ);
#pragma unroll
                            for (int j0 = 0; j0 < HQ / 4; j0 += 4) {           // four 16-byte reads in flight (all eight: spills)
                                f4 hv[4];
#pragma unroll
                                for (int j = 0; j < 4; j++) hv[j] = hb[j0 + j];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const osk::f2 lo = {hv[j][0], hv[j][1]}, hi = {hv[j][2], hv[j][3]};
                                    const int jj = 2 * (j0 + j);
                                    ar[0] = osk::fma2(wr[0][jj], lo, ar[0]); ar[1] = osk::fma2(wr[0][jj + 1], hi, ar[1]);
                                    az[0] = osk::fma2(wr[1][jj], lo, az[0]); az[1] = osk::fma2(wr[1][jj + 1], hi, az[1]);
                                    an[0] = osk::fma2(wr[2][jj], lo, an[0]); an[1] = osk::fma2(wr[2][jj + 1], hi, an[1]);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            const osk::f2 qr = ar[0] + ar[1], qz = az[0] + az[1], qn = an[0] + an[1];
                            sr = qr[0] + qr[1]; sz = qz[0] + qz[1]; sn = qn[0] + qn[1];
                            // the four column quarters of a unit are one DPP quad: quad_perm [1,0,3,2], then [2,3,0,1]
                            sr += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sr), 0xB1, 0xf, 0xf, true));
                            sz += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sz), 0xB1, 0xf, 0xf, true));
                            sn += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sn), 0xB1, 0xf, 0xf, true));
                            sr += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sr), 0x4E, 0xf, 0xf, true));
                            sz += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sz), 0x4E, 0xf, 0xf, true));
                            sn += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sn), 0x4E, 0xf, 0xf, true));
                        }
                        // cell update, the same on the four lanes of the quad; lane 0 of it writes
                        const float r = sigmoidf_(gir + sr), z = sigmoidf_(giz + sz);
                        const float nn = tanhf_(fmaf(r, sn + my_bhn, gin));
                        const float hn = fmaf(z, hprev[b] - nn, nn);                  // (1 - z) n + z h
                        hprev[b] = hn;
                        if (pc == 0) {
                            hn_buf[b * H + pu] = hn;
                            if (!last) xout[pu * NP + n] = hn;
                        }
                        __builtin_amdgcn_sched_barrier(0);           // one trajectory at a time (interleaved, their registers spill)
                    }
                }
            }
            VTS(4)                                                     // phase B: products, quad sums, cell update
            __syncthreads();
            VTS(5)
        }
        if (a.h_last && pb_active && pc == 0) {
#pragma unroll
            for (int b = 0; b < VEC_BMAX; b++)
                if (b < B) a.h_last[((size_t)l * B + b) * H + pu] = hprev[b];
        }
        w += vec_layer_floats(K, H);
    }
    // ---- head: fc (+ sigmoid) on the top layer's h_T ----
    const float *hcur = hbuf + (T & 1) * VEC_BMAX * H;
    for (int i = tid; i < B * a.C; i += VEC_THREADS) {
        const int b = i / a.C, c = i % a.C;
        float v0 = 0.f, v1 = 0.f;
        for (int u = 0; u < H; u += 2) {
            v0 = fmaf(a.fcw[(size_t)c * H + u], hcur[b * H + u], v0);
            v1 = fmaf(a.fcw[(size_t)c * H + u + 1], hcur[b * H + u + 1], v1);
        }
        const float v = (v0 + v1) + a.fcb[c];
        a.out[(size_t)b * a.C + c] = a.use_sigmoid ? sigmoidf_(v) : v;
    }
    VTS(7)                                                             // head
#ifdef OS_LAYER_TS
    if (tid == 0)
        printf("gru_vec_kernel<%d> B=%d T=%d L=%d cycles (thread 0): x stage %llu | phase A %llu | - %llu | barrier behind phase A %llu | phase B products + cell %llu | phase B barriers %llu | - %llu | head %llu | sum %llu\n",
               H, B, T, a.L, vts[0], vts[1], vts[2], vts[3], vts[4], vts[5], vts[6], vts[7], vts[0] + vts[1] + vts[2] + vts[3] + vts[4] + vts[5] + vts[6] + vts[7]);
#endif
}

}  // namespace osg

using namespace osg;

size_t os_layer_packed_floats(int K, int H)
{
    return (size_t)(H / 32) * chunk_floats((K + 1) / 2, H / 2);
}

int os_ensure_scratch(os_ctx *ctx, float **buf, size_t *cap, size_t need_floats)
{
    if (*cap >= need_floats) return 0;
    if (*buf) OS_HIP(ctx, hipFree(*buf));
    *buf = nullptr; *cap = 0;
    OS_HIP(ctx, hipMalloc((void **)buf, need_floats * sizeof(float)));
    *cap = need_floats;
    return 0;
}

extern "C" {

size_t os_gru_param_count(const os_gru_dims *d)
{
    size_t n = 0;
    for (int l = 0; l < d->num_layers; l++) {
        const size_t il = l == 0 ? d->input_size : d->hidden_size, H = d->hidden_size;
        n += 3 * H * il + 3 * H * H + 6 * H;
    }
    return n + (size_t)d->num_classes * d->hidden_size + d->num_classes;
}

static bool vec_dims_eligible(const os_gru_dims &d);

// key != 0: the caller's name for (these weights, in this state).  A slot holding the same key, dimensions and flat pointer is
// re-selected without packing; otherwise the least recently used of the four slots is (re)packed.  key == 0: always pack.
int os_gru_load_keyed(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, uint64_t key, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!d || !w_flat) return os_fail(ctx, -2, "os_gru_load: null pointer");
    const int H = d->hidden_size;
    if (H != 32 && H != 64 && H != 128) return os_fail(ctx, -4, "os_gru_load: hidden_size must be 32, 64 or 128");
    if (d->input_size <= 0 || d->num_layers <= 0 || d->num_layers > 16 || d->num_classes <= 0 || d->num_classes > 256)
        return os_fail(ctx, -4, "os_gru_load: unsupported dimensions");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_ctx::GruSlot *slot = nullptr;
    if (key)
        for (auto &sl : ctx->gru_slots)
            if (sl.packed && sl.key == key && sl.flat == w_flat && memcmp(&sl.d, d, sizeof(*d)) == 0) slot = &sl;
    const bool hit = slot != nullptr;
    if (!hit) {
        slot = &ctx->gru_slots[0];
        for (auto &sl : ctx->gru_slots)
            if (sl.stamp < slot->stamp) slot = &sl;           // an empty slot has stamp 0
        size_t total = 0;
        for (int l = 0; l < d->num_layers; l++) total += os_layer_packed_floats(l == 0 ? d->input_size : H, H);
        if (os_ensure_scratch(ctx, &slot->packed, &slot->cap, total)) return -10;
        // the B <= 4 kernel's transposed image is packed HERE too, by the same launch and so from the same snapshot of w_flat as
        // the MFMA image: a caller that edits w_flat in place without reloading keeps ONE consistent set of packed matrices on
        // every batch size (the head's fc weights are read from w_flat live on every path: documented in the header)
        const bool vec = vec_dims_eligible(*d);
        if (vec) {
            size_t vtotal = 0;
            for (int l = 0; l < d->num_layers; l++) vtotal += vec_layer_floats(l == 0 ? d->input_size : H, H);
            if (os_ensure_scratch(ctx, &slot->vec, &slot->vec_cap, vtotal)) return -10;
        }
        size_t src = 0, dst = 0, vdst = 0;
        PackAll pa;
        pa.n = d->num_layers; pa.H = H;
        for (int l = 0; l < d->num_layers; l++) {
            const int K = l == 0 ? d->input_size : H;
            pa.K[l] = K;
            pa.Wih[l] = w_flat + src; pa.Whh[l] = pa.Wih[l] + (size_t)3 * H * K; pa.bih[l] = pa.Whh[l] + (size_t)3 * H * H; pa.bhh[l] = pa.bih[l] + 3 * H;
            pa.dst[l] = slot->packed + dst;
            pa.vdst[l] = vec ? slot->vec + vdst : nullptr;
            src += (size_t)3 * H * K + (size_t)3 * H * H + 6 * (size_t)H;
            dst += os_layer_packed_floats(K, H);
            vdst += vec_layer_floats(K, H);
        }
        hipLaunchKernelGGL(gru_pack_all_kernel, dim3(H / 32, 16, (vec ? 2 : 1) * d->num_layers), dim3(256), 0, (hipStream_t)stream, pa);
        OS_HIP(ctx, hipGetLastError());
        slot->key = key; slot->d = *d; slot->flat = w_flat;
        slot->vec_valid = vec;
        slot->bf_spl = 0;                                       // the bf16 term image (if any) is of the previous weights
        ctx->gru_generation++;                                  // counts packs (os_gru_generation): a cache hit does not bump it
    }
    slot->stamp = ++ctx->gru_clock;
    ctx->gru_slot = slot;
    ctx->gru_packed = slot->packed;
    ctx->gru = *d;
    ctx->gru_flat = w_flat;
    ctx->gru_loaded = true;
    return 0;
}

int os_gru_load(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, void *stream)
{
    return os_gru_load_keyed(ctx, d, w_flat, 0, stream);
}

}  // extern "C"

// fc + sigmoid head on an SoA hidden state top [H][B]; fcw points at fc.weight followed by fc.bias.
// De-normalised prediction and error bands of gru/gru_test.py:184-189,208-213 in one pass: out [B][2 n] = [prediction (n) | error (n)]
// -> pred = p (max - min) + min, above = (p + e) (max - min) + min, below = (p - e) (max - min) + min, each [B][n].
__global__ void gru_bands_kernel(int B, int n, const float *__restrict__ out, const float *__restrict__ mn, const float *__restrict__ mx,
                                 float *__restrict__ pred, float *__restrict__ above, float *__restrict__ below)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    const int b = i / n, j = i - b * n;
    const float p = out[(size_t)b * 2 * n + j], e = out[(size_t)b * 2 * n + n + j], lo = mn[j], sc = mx[j] - mn[j];
    pred[i] = p * sc + lo;
    above[i] = (p + e) * sc + lo;
    below[i] = (p - e) * sc + lo;
}

int os_gru_head_launch(os_ctx *ctx, int B, const float *top, const float *fcw, float *out, hipStream_t s)
{
    const os_gru_dims &d = ctx->gru;
    const int H = d.hidden_size;
    const size_t hlds = ((size_t)8 * H + 8 + 3 * 8 * 64) * sizeof(float);
    const int hslot = os_prof_begin(ctx, OS_PHASE_GRU_HEAD, s, "gru_head_kernel");
    hipLaunchKernelGGL(gru_head_kernel, dim3((B + 63) / 64, (d.num_classes + 7) / 8), dim3(256), hlds, s, B, H, d.num_classes, top, fcw,
                       fcw + (size_t)d.num_classes * H, d.use_sigmoid, out);
    os_prof_end(ctx, hslot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

// LDS bytes of the eight-wave split body: h double buffer | x double buffer | exchange buffers.  H = 64 with K = 189..192 does not
// fit the 160 KB (round 4: found by a stack-kernel test; such a layer takes the plain kernel)
static size_t split_lds_bytes(int K, int H)
{
    const int NCH = H / 32, parts = 8 / NCH, KPx = (K + 1) / 2;
    return ((size_t)2 * 32 * (H + 1) + (size_t)2 * 32 * (2 * KPx + 1) + (size_t)NCH * (parts - 1) * 64 * 64) * sizeof(float);
}
// true when os_gru_launch_layer will pick gru_layer_ahead_kernel for this shape (the one kernel that can read a (B, T, K)
// batch_first input directly: LayerArgs.xs_btf)
static bool ahead_eligible(os_ctx *ctx, int B, int T, int K, int H)
{
    const bool split = H / 32 >= 2 && (B + 31) / 32 <= ctx->cu_count && K <= 192 && ctx->tune_gru_split != 0;
    return split && H == 128 && (size_t)T * B * H * 4 < ((size_t)1 << 32) && (size_t)T * B * K * 4 < ((size_t)1 << 32) &&
           ctx->tune_gru_ahead != 0;
}
// Layers per pipelined launch (gru_stack_kernel: the eight-wave split body) for a stack of nlayers on this batch: as many as leave every
// (layer, tile) workgroup of a launch resident at once, at most eight; 0 = a launch per layer.  All of them up to 2,048 trajectories at
// four layers; two at a time up to 4,096 (the reference's own data set: ~4,050 windows, gru/gru_test.py:138-140).
static int stack_group(os_ctx *ctx, int B, int T, int Kfirst, int H, int nlayers)
{
    const int NCH = H / 32, tiles = (B + 31) / 32;
    if (!(ctx->tune_gru_stack != 0 && nlayers >= 2 && (NCH == 4 || NCH == 2 || NCH == 1) && Kfirst <= 192 && H <= 192 &&
          split_lds_bytes(Kfirst, H) <= 160 * 1024 && (size_t)T * B * (Kfirst > H ? Kfirst : H) * 4 < ((size_t)1 << 31)))
        return 0;
    int g = ctx->cu_count / tiles;
    g = g > 8 ? 8 : g;
    g = g > nlayers ? nlayers : g;
    return g >= 2 ? g : 0;
}
// true when the WHOLE stack is one launch
static bool stack_eligible(os_ctx *ctx, int B, int T, int Kfirst, int H, int nlayers)
{
    return nlayers <= 8 && stack_group(ctx, B, T, Kfirst, H, nlayers) == nlayers;
}
bool os_gru_layer_takes_btf(os_ctx *ctx, int B, int T, int K, int H)
{
    if (ctx->gru_split_bf16 && H == 128) return false;      // the opt-in bf16 layer kernel reads the SoA stream
    return ahead_eligible(ctx, B, T, K, H);
}

bool os_gru_stack_eligible(os_ctx *ctx, int B, int T, int Kfirst, int H, int nlayers) { return stack_eligible(ctx, B, T, Kfirst, H, nlayers); }

// One gru_stack_kernel launch over n <= 8 consecutive layers (layer i + 1 reads layer i's seq_out); shared by inference and the
// training forward (saved activations and every layer's sequence come out as from the per-layer kernels).
int os_gru_launch_stack(os_ctx *ctx, const LayerArgs *layers, int n, hipStream_t s)
{
    const int B = layers[0].B, H = layers[0].H, NCH = H / 32;
    StackArgs sa;
    sa.n = n; sa.tiles = (B + 31) / 32;
    const size_t nfl = (size_t)sa.n * sa.tiles;
    if (ctx->stack_flags_n < nfl) {
        if (ctx->stack_flags) OS_HIP(ctx, hipFree(ctx->stack_flags));
        ctx->stack_flags = nullptr; ctx->stack_flags_n = 0;
        OS_HIP(ctx, hipMalloc((void **)&ctx->stack_flags, nfl * sizeof(uint32_t)));
        ctx->stack_flags_n = nfl;
    }
    sa.flags = ctx->stack_flags;
    sa.err = ctx->stack_err_dev; sa.err_local = ctx->stack_err_local; sa.max_polls = ctx->stack_max_polls;
    sa.drop_layer = ctx->stack_dbg_drop_layer; sa.drop_step = ctx->stack_dbg_drop_step;
    OS_HIP(ctx, hipMemsetAsync(sa.flags, 0, nfl * sizeof(uint32_t), s));
    size_t lds_max = 0;
    for (int l = 0; l < n; l++) {
        sa.layer[l] = layers[l];
        const size_t lds_l = split_lds_bytes(layers[l].K, H);
        lds_max = lds_l > lds_max ? lds_l : lds_max;
    }
    if (NCH != 4 && NCH != 2 && NCH != 1) return os_fail(ctx, -4, "os_gru_launch_stack: hidden_size must be 128, 64 or 32");
    bool save = layers[0].sv_r != nullptr;
    for (int l = 1; l < n; l++)
        if ((layers[l].sv_r != nullptr) != save) return os_fail(ctx, -4, "os_gru_launch_stack: every layer of a launch saves its activations or none does");
    if (!ctx->stack_attr_set) {
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_stack_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->stack_attr_set = true;
    }
    const int slot = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, "gru_stack_kernel");
    const dim3 grid(sa.tiles, sa.n), block(512);
    if (NCH == 4) { if (save) hipLaunchKernelGGL((gru_stack_kernel<4, true>), grid, block, lds_max, s, sa); else hipLaunchKernelGGL((gru_stack_kernel<4, false>), grid, block, lds_max, s, sa); }
    else if (NCH == 2) { if (save) hipLaunchKernelGGL((gru_stack_kernel<2, true>), grid, block, lds_max, s, sa); else hipLaunchKernelGGL((gru_stack_kernel<2, false>), grid, block, lds_max, s, sa); }
    else { if (save) hipLaunchKernelGGL((gru_stack_kernel<1, true>), grid, block, lds_max, s, sa); else hipLaunchKernelGGL((gru_stack_kernel<1, false>), grid, block, lds_max, s, sa); }      // (H = 32, eight slices per chunk: outside the scratch-free contract)
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return os_stack_verify(ctx, s, "gru_stack_kernel");
}

// After a launch of a progress-counter kernel.  OS_GRU_STACK=1 (default): wait for it and read the error word -- a consumer whose
// bounded wait expired (its producer never became resident, or another process held the producer's CUs for seconds) makes THIS call
// fail with -20; the caller re-runs it with a launch per layer (os_gru_set_stack(ctx, 0): the Python engine does).  The wait costs
// ~10 us on launches of 0.2-2 ms that their callers read back at once.  OS_GRU_STACK=2: no wait; the word is looked at by
// os_stack_pending at the entry of the next os_gru_* call, and adam_kernel keeps the weights unchanged while it is set.
int os_stack_verify(os_ctx *ctx, hipStream_t s, const char *what)
{
    if (ctx->tune_gru_stack != 1) { ctx->stack_dirty = true; return 0; }      // asynchronous: os_stack_check / os_stack_pending look later
    OS_HIP(ctx, hipStreamSynchronize(s));
    if (*(volatile int32_t *)ctx->stack_err_host) {
        *(volatile int32_t *)ctx->stack_err_host = 0;
        (void)hipMemsetAsync(ctx->stack_err_local, 0, sizeof(int32_t), s);      // (the stream is drained: nothing queued reads the twin)
        char msg[256];
        snprintf(msg, sizeof(msg), "%s: a layer's bounded wait for the layer before it expired (progress-counter pipeline); outputs are NaN-poisoned -- "
                 "re-run with a launch per layer (os_gru_set_stack(ctx, 0) / OS_GRU_STACK=0)", what);
        return os_fail(ctx, -20, msg);
    }
    return 0;
}
// Entry of later calls (mode 2).  The device twin is what a queued adam_kernel reads: it may only be cleared once everything enqueued
// behind the lost producer has run, on WHATEVER stream (a caller's non-blocking stream does not order against the null stream), so
// the report path drains the device first.  An error path: the cost does not matter.
int os_stack_pending(os_ctx *ctx, const char *what)
{
    if (*(volatile int32_t *)ctx->stack_err_host) {
        (void)hipDeviceSynchronize();
        *(volatile int32_t *)ctx->stack_err_host = 0;
        (void)hipMemset(ctx->stack_err_local, 0, sizeof(int32_t));
        ctx->stack_dirty = false;
        char msg[256];
        snprintf(msg, sizeof(msg), "%s: an EARLIER stacked launch on this context lost a producer (its outputs were NaN-poisoned, optimiser steps behind it "
                 "were skipped): re-run that work with os_gru_set_stack(ctx, 0)", what);
        return os_fail(ctx, -20, msg);
    }
    return 0;
}
extern "C" int os_stack_check(os_ctx *ctx, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (!ctx->stack_dirty) return 0;                       // no asynchronous stacked launch since the last check: nothing to wait for
    OS_HIP(ctx, hipSetDevice(ctx->device));
    OS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    ctx->stack_dirty = false;
    if (*(volatile int32_t *)ctx->stack_err_host) {
        *(volatile int32_t *)ctx->stack_err_host = 0;
        (void)hipMemsetAsync(ctx->stack_err_local, 0, sizeof(int32_t), (hipStream_t)stream);
        return os_fail(ctx, -20, "os_stack_check: a stacked launch since the last check lost a producer; what it and the launches behind it wrote is "
                                 "NaN-poisoned -- redo that work with os_gru_set_stack(ctx, 0)");
    }
    return 0;
}

// Launches gru_layer_kernel for one layer (shared by inference and the training forward).
int os_gru_launch_layer(os_ctx *ctx, const LayerArgs &a, hipStream_t s)
{
    if (ctx->gru_split_bf16) {               // opt-in reduced precision: H = 128 inference layers at large batches (gru_bf16_kernels.hip)
        bool done = false;
        const int rc = os_gru_try_layer_bf16(ctx, a, s, &done);
        if (rc || done) return rc;
    }
    const int H = a.H, NCH = H / 32;
    const int WPC = NCH >= 4 ? 1 : 4 / NCH;
    // rows per workgroup: H = 64: RBW 2 x 2 waves/chunk = 128; H = 128: RBW 2 = 64; H = 32: RBW 1 x 4 waves/chunk = 128
    int RBW = H == 32 ? 1 : 2;
    // small batches: halve the row tile when the grid would not put two workgroups on every compute unit (256 CUs; the
    // recurrence cannot be split across workgroups, so rows are the only parallel axis)
    if (RBW == 2 && (a.B + 32 * RBW * WPC - 1) / (32 * RBW * WPC) < 2 * ctx->cu_count) RBW = 1;
    const int BM = 32 * RBW * WPC;
    const size_t lds = (size_t)2 * BM * (H + 1) * sizeof(float);      // double-buffered h tile (<= 66.6 KB)
    if (!ctx->layer_attr_set) {            // per context (= per device)
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        ctx->layer_attr_set = true;
    }
    dim3 grid((a.B + BM - 1) / BM), block(256);
    // at most one 32-row tile per CU: eight waves on one tile (slices of the gate GEMM's reduction);
    // measured (60,128,4), T = 100: B = 4096 35 -> 48, B = 8192 70 -> 92 TFLOP/s; from two tiles per CU on the plain kernel wins
    bool split = NCH >= 2 && (a.B + 31) / 32 <= ctx->cu_count;       // H = 32 (eight slices per chunk) loses: plain kernel
    if (a.K > 192 || split_lds_bytes(a.K, H) > 160 * 1024) split = false;   // its x tile staging covers 12 x 16 inputs
    if (ctx->tune_gru_split == 0) split = false;
    const bool ahead = ahead_eligible(ctx, a.B, a.T, a.K, H);
    if (a.xs_btf && !ahead) return os_fail(ctx, -4, "os_gru_launch_layer: a batch_first input needs the ahead kernel's shape");
    // large batches, H = 128 / 64, inference: the x tile travels global -> LDS by DMA one step ahead (gru_layer_stage_kernel)
    const int bm_st = STAGE_BM * (NCH >= 4 ? 1 : 2);
    const size_t lds_st = ((size_t)bm_st * (H + 1) + (size_t)((a.K + 3) & ~3) * bm_st) * sizeof(float);
    const bool stage = !ahead && !split && RBW == 2 && (H == 128 || H == 64) && a.KPx >= 2 * STAGE_DEPTH && !a.sv_r && !a.xs_btf && a.B % 4 == 0 &&
                       lds_st <= 80 * 1024 && (size_t)a.K * a.B * 4 < ((size_t)1 << 31) && ctx->tune_gru_stage != 0;
    const int slot = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s,
                                   ahead ? "gru_layer_ahead_kernel" : split ? "gru_layer_split_kernel" : stage ? "gru_layer_stage_kernel"
                                   : (RBW == 2 ? "gru_layer_kernel<2,2>" : "gru_layer_kernel<1,3>"));
    if (stage) {
        if (!ctx->stage_attr_set) {
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_stage_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            ctx->stage_attr_set = true;
        }
        if (H == 128) hipLaunchKernelGGL(gru_layer_stage_kernel<4>, dim3((a.B + bm_st - 1) / bm_st), dim3(256), lds_st, s, a);
        else hipLaunchKernelGGL(gru_layer_stage_kernel<2>, dim3((a.B + bm_st - 1) / bm_st), dim3(256), lds_st, s, a);
    } else if (ahead) {
        const size_t lds_a = ((size_t)2 * 32 * (H + 1) + (size_t)32 * (2 * a.KPx + 1) + (size_t)4 * 48 * 64) * sizeof(float);
        if (!ctx->ahead_attr_set) {
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_ahead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->ahead_attr_set = true;
        }
        hipLaunchKernelGGL(gru_layer_ahead_kernel, dim3((a.B + 31) / 32), dim3(512), lds_a, s, a);
    } else if (split) {
        const int parts = 8 / NCH;
        const size_t lds_s = ((size_t)2 * 32 * (H + 1) + (size_t)2 * 32 * (2 * a.KPx + 1) + (size_t)NCH * (parts - 1) * 64 * 64) * sizeof(float);
        if (!ctx->split_attr_set) {
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->split_attr_set = true;
        }
        const dim3 g32((a.B + 31) / 32), b512(512);
        const bool save = a.sv_r != nullptr;
        if (NCH == 4) { if (save) hipLaunchKernelGGL((gru_layer_split_kernel<4, true>), g32, b512, lds_s, s, a); else hipLaunchKernelGGL((gru_layer_split_kernel<4, false>), g32, b512, lds_s, s, a); }
        else { if (save) hipLaunchKernelGGL((gru_layer_split_kernel<2, true>), g32, b512, lds_s, s, a); else hipLaunchKernelGGL((gru_layer_split_kernel<2, false>), g32, b512, lds_s, s, a); }
    } else if (RBW == 2) hipLaunchKernelGGL((gru_layer_kernel<2, 2>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gru_layer_kernel<1, 3>), grid, block, lds, s, a);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

// Scratch for the layer stack: two ping-pong sequence buffers [T][H][B] (only for L > 1) and one h_last [H][B].
int os_gru_scratch(os_ctx *ctx, int B, int T, float **seq0, float **seq1, float **hlast)
{
    const os_gru_dims &d = ctx->gru;
    const size_t seqf = (size_t)T * d.hidden_size * B, hf = (size_t)d.hidden_size * B;
    const size_t need = (d.num_layers > 1 ? 2 * seqf : 0) + hf;
    if (os_ensure_scratch(ctx, &ctx->gru_seq, &ctx->gru_seq_floats, need)) return -10;
    *seq0 = ctx->gru_seq; *seq1 = ctx->gru_seq + seqf;
    *hlast = ctx->gru_seq + (d.num_layers > 1 ? 2 * seqf : 0);
    return 0;
}

// Runs layers first_layer..L-1 and the head.  `in` is the SoA input of layer first_layer ([T][K][B]); for
// first_layer > 0 it must be the ping-pong buffer seq[(first_layer-1)&1] returned by os_gru_scratch.
static int gru_layers(os_ctx *ctx, int B, int T, const float *in, int in_btf, int first_layer, float *out, float *h_last_all,
                      hipStream_t s)
{
    const os_gru_dims &d = ctx->gru;
    const int H = d.hidden_size, L = d.num_layers, NCH = H / 32;
    float *seqbuf[2], *hlast;
    if (os_gru_scratch(ctx, B, T, &seqbuf[0], &seqbuf[1], &hlast)) return -10;
    const size_t hf = (size_t)H * B;
    size_t woff = 0;
    for (int l = 0; l < first_layer; l++) woff += os_layer_packed_floats(l == 0 ? d.input_size : H, H);
    // ---- small H = 128 batch: every layer in ONE launch on four CUs per (layer, tile) (gru_wide_kernel.hip) ----
    if (first_layer == 0 && L >= 2 && os_gru_wide_eligible(ctx, B, T, d.input_size, H, L)) {
        const size_t tbh = (size_t)T * B * H;
        if (os_ensure_scratch(ctx, &ctx->gru_wide_seq, &ctx->gru_wide_seq_floats, (size_t)L * tbh)) return -10;
        WideArgs wa;
        wa.n = L; wa.tiles = (B + 31) / 32; wa.B = B; wa.T = T; wa.K0 = d.input_size; wa.xs0 = in; wa.xs0_btf = in_btf;
        for (int l = 0; l < L; l++) {
            wa.w[l] = ctx->gru_packed + woff;
            wa.hseq[l] = ctx->gru_wide_seq + (size_t)l * tbh;
            wa.sv_r[l] = wa.sv_z[l] = wa.sv_n[l] = wa.sv_g[l] = nullptr;
            wa.h_last[l] = h_last_all ? h_last_all + (size_t)l * hf : ((l == L - 1) ? hlast : nullptr);
            woff += os_layer_packed_floats(l == 0 ? d.input_size : H, H);
        }
        if (const int rc = os_gru_launch_wide(ctx, wa, false, s)) return rc;
        const float *top = h_last_all ? h_last_all + (size_t)(L - 1) * hf : hlast;
        const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
        return os_gru_head_launch(ctx, B, top, fcw, out, s);
    }
    const int grp = in_btf ? 0 : stack_group(ctx, B, T, first_layer == 0 ? d.input_size : H, H, L - first_layer);
    if (grp >= 2) {
        // ---- small batch: grp layers per launch, pipelined through progress flags (gru_stack_kernel); a single left-over layer on its own ----
        const float *lin = in;
        for (int l0 = first_layer; l0 < L; l0 += grp) {
            const int n = L - l0 < grp ? L - l0 : grp;
            LayerArgs la[8];
            for (int l = l0; l < l0 + n; l++) {
                const int K = l == 0 ? d.input_size : H;
                LayerArgs &a = la[l - l0];
                a.B = B; a.T = T; a.K = K; a.H = H; a.KPx = (K + 1) / 2; a.KPh = H / 2;
                a.xs = lin; a.xs_btf = 0; a.w = ctx->gru_packed + woff;
                a.seq_out = (l < L - 1) ? seqbuf[l & 1] : nullptr;
                a.h_last = h_last_all ? h_last_all + (size_t)l * hf : ((l == L - 1) ? hlast : nullptr);
                a.sv_r = a.sv_z = a.sv_n = a.sv_g = a.sv_h = nullptr;
                lin = a.seq_out;
                woff += os_layer_packed_floats(K, H);
            }
            const int rc = n >= 2 ? os_gru_launch_stack(ctx, la, n, s) : os_gru_launch_layer(ctx, la[0], s);
            if (rc) return rc;
        }
        const float *top = h_last_all ? h_last_all + (size_t)(L - 1) * hf : hlast;
        const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
        return os_gru_head_launch(ctx, B, top, fcw, out, s);
    }
    for (int l = first_layer; l < L; l++) {
        const int K = l == 0 ? d.input_size : H;
        LayerArgs a;
        a.B = B; a.T = T; a.K = K; a.H = H; a.KPx = (K + 1) / 2; a.KPh = H / 2;
        a.xs = in; a.xs_btf = (l == first_layer) ? in_btf : 0; a.w = ctx->gru_packed + woff;
        a.seq_out = (l < L - 1) ? seqbuf[l & 1] : nullptr;
        a.h_last = h_last_all ? h_last_all + (size_t)l * hf : ((l == L - 1) ? hlast : nullptr);
        a.sv_r = a.sv_z = a.sv_n = a.sv_g = a.sv_h = nullptr;
        if (os_gru_launch_layer(ctx, a, s)) return -10;
        in = a.seq_out;
        woff += os_layer_packed_floats(K, H);
    }
    const float *top = h_last_all ? h_last_all + (size_t)(L - 1) * hf : hlast;
    const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
    return os_gru_head_launch(ctx, B, top, fcw, out, s);
}

int os_gru_layers_impl(os_ctx *ctx, int B, int T, const float *in, int first_layer, float *out, float *h_last_all, hipStream_t s)
{
    return gru_layers(ctx, B, T, in, 0, first_layer, out, h_last_all, s);
}

extern "C" {

// Runs the layer stack on an SoA input sequence xs [T][I][B] (device).
int os_gru_forward_soa(os_ctx *ctx, int32_t B, int32_t T, const float *xs, float *out, float *h_last_all, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (int rcp = os_stack_pending(ctx, "os_gru_forward_soa")) return rcp;
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_gru_forward: call os_gru_load first");
    if (B <= 0 || T <= 0 || !xs || !out) return os_fail(ctx, -2, "os_gru_forward: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    return os_gru_layers_impl(ctx, B, T, xs, 0, out, h_last_all, (hipStream_t)stream);
}

// B <= 4 windows of at most 48 / B steps at the reference's widths: the single-workgroup vector kernel
// LDS the single-workgroup kernel needs for B windows of T steps (x tile | h sequence | gi | h double buffer)
static size_t vec_lds_bytes(const os_gru_dims &d, int B, int T)
{
    const int H = d.hidden_size, R = 3 * H;
    const int NP = (B * T + 3) & ~3, KA = d.input_size > H ? d.input_size : H;
    return ((size_t)KA * NP + (size_t)H * NP + (size_t)NP * R + (size_t)2 * VEC_BMAX * H) * sizeof(float);
}
static bool vec_dims_eligible(const os_gru_dims &d)
{
    return (d.hidden_size == 128 || d.hidden_size == 64) && d.input_size <= 192;
}
static bool vec_eligible(os_ctx *ctx, int B, int T)
{
    const os_gru_dims &d = ctx->gru;
    return ctx->tune_gru_vec != 0 && B <= VEC_BMAX && B * T <= VEC_NMAX && vec_dims_eligible(d) &&
           vec_lds_bytes(d, B, T) <= (size_t)160 * 1024;          // computed, not implied by the N / K caps
}
static int gru_vec_launch(os_ctx *ctx, int B, int T, const float *x, float *out, float *h_last, hipStream_t s)
{
    const os_gru_dims &d = ctx->gru;
    const int H = d.hidden_size, L = d.num_layers;
    os_ctx::GruSlot *slot = ctx->gru_slot;
    if (!slot->vec_valid) return os_fail(ctx, -5, "gru_vec_launch: no packed image (os_gru_load packs it for eligible dimensions)");
    VecArgs a;
    a.B = B; a.T = T; a.K0 = d.input_size; a.L = L; a.C = d.num_classes; a.use_sigmoid = d.use_sigmoid;
    a.x = x; a.wvec = slot->vec;
    a.fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
    a.fcb = a.fcw + (size_t)d.num_classes * H;
    a.out = out; a.h_last = h_last;
    const size_t lds = vec_lds_bytes(d, B, T);
    if (!ctx->vec_attr_set) {
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_vec_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_vec_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->vec_attr_set = true;
    }
    const int slot_p = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, "gru_vec_kernel");
    if (H == 128) hipLaunchKernelGGL(gru_vec_kernel<128>, dim3(1), dim3(VEC_THREADS), lds, s, a);
    else hipLaunchKernelGGL(gru_vec_kernel<64>, dim3(1), dim3(VEC_THREADS), lds, s, a);
    os_prof_end(ctx, slot_p, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

// gru/gru_test.py:138-140,174-191: the reference builds overlapping windows from ONE row stream (window i = rows i .. i + W - 1) and
// evaluates them one by one.  Here all N - W + 1 windows run as one batch WITHOUT being materialised (the windowed tensor is W
// times the stream) and the input half of layer 0's gate GEMM is computed once per ROW (gru_gi_kernel) instead of once per
// (window, step): rows [N][I] row-major on the device -> out [N - W + 1][C].
int os_gru_set_stack(os_ctx *ctx, int32_t mode)
{
    OS_CHECK_CTX(ctx);
    if (mode < 0 || mode > 2) return os_fail(ctx, -2, "os_gru_set_stack: 0 (a launch per layer), 1 (stacked, verified) or 2 (stacked, asynchronous)");
    ctx->tune_gru_stack = mode;
    return 0;
}
int os_gru_get_stack(const os_ctx *ctx) { return (ctx && ctx->magic == OS_MAGIC) ? ctx->tune_gru_stack : -1; }

int os_gru_forward_windows(os_ctx *ctx, int32_t N, int32_t W, const float *rows, float *out, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (int rcp = os_stack_pending(ctx, "os_gru_forward_windows")) return rcp;
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_gru_forward_windows: call os_gru_load first");
    if (N <= 0 || W <= 0 || W > N || !rows || !out) return os_fail(ctx, -2, "os_gru_forward_windows: bad argument (1 <= window <= rows)");
    const os_gru_dims &d = ctx->gru;
    const int I = d.input_size, H = d.hidden_size, L = d.num_layers, NCH = H / 32, B = N - W + 1, T = W;
    if ((H != 128 && H != 64) || I > 192 || split_lds_bytes(I, H) > 160 * 1024)
        return os_fail(ctx, -4, "os_gru_forward_windows: hidden_size 128 or 64 and input_size <= 192 (materialise the windows and call os_gru_forward otherwise)");
    if ((size_t)N * 3 * H * 4 >= ((size_t)1 << 32) || (size_t)T * B * H * 4 >= ((size_t)1 << 31))
        return os_fail(ctx, -2, "os_gru_forward_windows: stream too long for 32-bit buffer offsets");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (os_ensure_scratch(ctx, &ctx->gru_gi, &ctx->gru_gi_floats, (size_t)N * 3 * H)) return -10;
    float *seqbuf[2], *hlast;
    if (os_gru_scratch(ctx, B, T, &seqbuf[0], &seqbuf[1], &hlast)) return -10;
    // ---- gi = rows . W_ih(layer 0)^T, once per row ----
    GiArgs g;
    g.N = N; g.K = I; g.H = H; g.KPx = (I + 1) / 2; g.KPh = H / 2; g.rows = rows; g.w = ctx->gru_packed; g.gi = ctx->gru_gi;
    {
        const int rbw = (N + 63) / 64 >= 2 * ctx->cu_count ? 2 : 1;
        const size_t lds_g = (size_t)32 * rbw * (2 * g.KPx + 1) * sizeof(float);
        const int slot = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, "gru_gi_kernel");
        if (rbw == 2) hipLaunchKernelGGL(gru_gi_kernel<2>, dim3((N + 63) / 64), dim3(256), lds_g, s, g);
        else hipLaunchKernelGGL(gru_gi_kernel<1>, dim3((N + 31) / 32), dim3(256), lds_g, s, g);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
    }
    // ---- layer 0 on the windows: recurrent half + gi tile ----
    LayerArgs a;
    a.B = B; a.T = T; a.K = I; a.H = H; a.KPx = (I + 1) / 2; a.KPh = H / 2;
    a.xs = nullptr; a.xs_btf = 0; a.w = ctx->gru_packed;
    a.seq_out = L > 1 ? seqbuf[0] : nullptr;
    a.h_last = L == 1 ? hlast : nullptr;
    a.sv_r = a.sv_z = a.sv_n = a.sv_g = a.sv_h = nullptr;
    a.gi = ctx->gru_gi; a.gi_rows = N;
    {
        const size_t lds_s = split_lds_bytes(I, H);
        if (!ctx->gi_attr_set) {
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_split_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->gi_attr_set = true;
        }
        const int slot = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, "gru_layer_split_kernel<GI>");
        const dim3 g32((B + 31) / 32), b512(512);
        if (NCH == 4) hipLaunchKernelGGL((gru_layer_split_kernel<4, false, true>), g32, b512, lds_s, s, a);
        else hipLaunchKernelGGL((gru_layer_split_kernel<2, false, true>), g32, b512, lds_s, s, a);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
    }
    if (L == 1) {
        const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * H + d.num_classes));
        return os_gru_head_launch(ctx, B, hlast, fcw, out, s);
    }
    return gru_layers(ctx, B, T, seqbuf[0], 0, 1, out, nullptr, s);
}

int os_gru_bands(os_ctx *ctx, int32_t B, int32_t n, const float *out, const float *min_v, const float *max_v, float *pred, float *above,
                 float *below, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || n <= 0 || !out || !min_v || !max_v || !pred || !above || !below) return os_fail(ctx, -2, "os_gru_bands: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gru_bands_kernel, dim3((B * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n, out, min_v, max_v, pred, above, below);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

int os_gru_forward(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, float *h_last, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (int rcp = os_stack_pending(ctx, "os_gru_forward")) return rcp;
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_gru_forward: call os_gru_load first");
    if (B <= 0 || T <= 0 || !x || !out) return os_fail(ctx, -2, "os_gru_forward: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    // (B, T, I) batch_first as the reference passes it -> SoA [T][I][B] in context scratch, unless the first layer's kernel
    // reads that layout itself (small batches at H = 128)
    const int I = ctx->gru.input_size;
    const int H = ctx->gru.hidden_size, L = ctx->gru.num_layers;
    if (vec_eligible(ctx, B, T)) return gru_vec_launch(ctx, B, T, x, out, h_last, (hipStream_t)stream);
    // (the stack kernel reads an SoA first layer: packing costs a few microseconds at its sizes)
    const bool wide = L >= 2 && os_gru_wide_eligible(ctx, B, T, I, H, L);      // gru_wide_kernel reads the (B, T, I) tensor itself
    const bool x_direct = wide || (stack_group(ctx, B, T, I, H, L) < 2 && os_gru_layer_takes_btf(ctx, B, T, I, H));
    int rc = 0;
    if (!x_direct) {
        if (os_ensure_scratch(ctx, &ctx->gru_xs, &ctx->gru_xs_floats, (size_t)B * T * I)) return -10;
        rc = os_pack_stream(ctx, B, T, I, x, ctx->gru_xs, stream);
        if (rc) return rc;
    }
    // h_last is requested in torch layout [L][B][H]; produce SoA [L][H][B] then transpose
    float *hl_soa = nullptr;
    if (h_last) {
        if (os_ensure_scratch(ctx, &ctx->gru_hl, &ctx->gru_hl_floats, (size_t)L * H * B)) return -10;
        hl_soa = ctx->gru_hl;
    }
    rc = x_direct ? gru_layers(ctx, B, T, x, 1, 0, out, hl_soa, (hipStream_t)stream) : os_gru_forward_soa(ctx, B, T, ctx->gru_xs, out, hl_soa, stream);
    if (rc) return rc;
    if (h_last)
        for (int l = 0; l < L; l++) {
            rc = os_unpack_stream(ctx, B, 1, H, hl_soa + (size_t)l * H * B, h_last + (size_t)l * B * H, stream);
            if (rc) return rc;
        }
    return 0;
}

}  // extern "C"
