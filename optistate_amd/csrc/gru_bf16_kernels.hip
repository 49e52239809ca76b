// gru_bf16_kernels.hip -- OPT-IN reduced-precision gate GEMM for the H = 128 GRU layers at large batches (os_gru_set_split_bf16 /
// OS_GRU_SPLIT_BF16 = 3 | 2; never the default: the default layer kernels are exact fp32).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles for 2 k) runs at 1/16 of the bf16 rate on gfx950 (v_mfma_f32_32x32x16_bf16: 32 cycles
// for 16 k).  gru_layer_stage_kernel sits at 0.87 of the fp32 peak; the only way past it is fewer matrix cycles.  Here every fp32
// operand of the gate GEMM is split into SPL bf16 terms (v = hi + mid + lo, 8 significant bits each; gru_common.hpp split_pair) and
// the products are formed on the bf16 MFMA with fp32 accumulation: SPL = 3 issues hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid (every
// term down to 2^-16 of a product, what is dropped is below fp32 rounding), SPL = 2 issues hi.hi, hi.lo, lo.hi (relative 2^-16 per
// product).  Per (64-trajectory tile, chunk, step) 2 x 20 x 18 = 720 (360) instructions of 32 cycles instead of 2 x 158 x 6 of 64.
//
// Shape of the kernel (a sibling of gru_layer_stage_kernel, whose skeleton it keeps: x tile by LDS-DMA a step ahead, h in LDS,
// two barriers per step, the fp32 cell update on the accumulator layout):
//   * TWO independent four-wave workgroups per CU, each on a 64-trajectory tile (xS [K4][64] | hS [64][129], both fp32: exactly
//     the stage kernel's images, 81,152 B at K = 188), a wave = one 32-column chunk of r, z, n x two 32-row blocks, 256 registers:
//     128 fp32 accumulators, the bf16 weight terms of two k-blocks (9 or 6 x 16 bytes per lane each, straight from L2 in fragment
//     order, requested one k-block ahead), ONE set of fp32 A values (read from LDS) and of the fragments they are split into on
//     the VALU.  No software pipelining inside a wave: the SIMD's other wave belongs to the other workgroup, drifts out of phase
//     with this one, and its MFMAs cover this wave's LDS reads, operand splits, waits and cell update;
//   * k runs over 16-wide blocks: KBx of the x part (K rounded up to 16, and to an even block count: zero weights; the A values
//     of k >= K are re-reads of the trajectory's own last input, so nothing foreign -- and no NaN of another row -- enters) and
//     8 of the h part;
//   * h_t leaves for seq_out one step LATE, from hS, inside step t + 1's k loop (256-byte runs; per-lane 16-byte pieces out of
//     the cell update cost 4.5 k exposed cycles per step in the first version).
// What bounds it (profiles/r05_pmc_bf16_layer.txt): POWER.  Under this load the chip drops its clock: 1.9 GHz at 60 % matrix-pipe
// occupancy (first version: one 4-wave workgroup per CU on a 128-row tile, software-pipelined splits), 1.7 GHz at 68 % (this one):
// 11 % fewer cycles became 2 % less time with three terms, 8 % with two.  The fp32 kernel holds 2.3 GHz at 87 %.
#include "launch.hpp"

#include "gru_common.hpp"
#include "gru_device.hpp"
#include "kf_device.hpp"

namespace osg {

using osk::make_rsrc;
using osk::rsrc_t;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BF_BM = 64, BF_H = 128, BF_HS = 129, BF_KBH = BF_H / 16;

__host__ __device__ inline int bf_kbx(int K) { return (((K + 15) / 16) + 1) & ~1; }                        // even
__host__ __device__ inline size_t bf_layer_dwords(int K, int spl) { return (size_t)4 * (bf_kbx(K) + BF_KBH) * 3 * spl * 256; }

// fp32 fragment image of a layer (gru_pack_all_kernel) -> bf16 term image [chunk][k-block][gate][term][64 lanes][4 dwords]:
// lane (li, lh) holds column chunk * 32 + li of the gate, k = 16 kb + 8 lh + 0..7 as four dwords of (even k, odd k) pairs
template <int SPL>
__global__ void gru_pack_bf16_kernel(const float *__restrict__ img, int KPx, int KPh, int KBx, uint32_t *__restrict__ dst)
{
    const int KB = KBx + KPh / 8;
    const int total = 4 * KB * 3 * 64 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int r = i;
        const int jj = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int g = r % 3; r /= 3;
        const int kb = r % KB, c = r / KB;
        const int li = lane & 31, lh = lane >> 5;
        const float *wx = img + (size_t)c * chunk_floats(KPx, KPh), *wh = wx + (size_t)KPx * 3 * 64;
        float w[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (kb < KBx) {
                const int k = 16 * kb + 8 * lh + 2 * jj + q, kp = k >> 1;
                w[q] = kp < KPx ? wx[(kp * 3 + g) * 64 + (k & 1) * 32 + li] : 0.f;        // (k >= K inside the last pair is zero in img)
            } else {
                const int k = 16 * (kb - KBx) + 8 * lh + 2 * jj + q, kp = k >> 1;
                w[q] = wh[(kp * 3 + g) * 64 + (k & 1) * 32 + li];
            }
        }
        uint32_t t[SPL];
        split_pair<SPL>(w[0], w[1], t);
#pragma unroll
        for (int sp = 0; sp < SPL; sp++) dst[((((size_t)c * KB + kb) * 3 + g) * SPL + sp) * 256 + lane * 4 + jj] = t[sp];
    }
}

// (non-template helpers: see gru_device.hpp)
__device__ __forceinline__ u32x4 buf_load4(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A values of one row block of a k-block: k = 8 lh .. 8 lh + 7 as two quads
struct Raw8 { f32x4 q[2]; };

// eight fp32 values -> SPL A fragments of bf16 terms
template <int SPL>
__device__ __forceinline__ void split8(const Raw8 &v, u32x4 *F)
{
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        uint32_t t[SPL];
        split_pair<SPL>(v.q[jj >> 1][2 * (jj & 1)], v.q[jj >> 1][2 * (jj & 1) + 1], t);
#pragma unroll
        for (int sp = 0; sp < SPL; sp++) F[sp][jj] = t[sp];
    }
}

// Every LDS access inside the T loop is inline assembly (an LDS-DMA is in the loop: gru_device.hpp); immediates fold (j, rb).
// x part: xS [k][128]: the eight k of a lane are 512 bytes apart -- eight ds_read_b32
template <int OFF, int JS>
__device__ __forceinline__ void read8(uint32_t addr, Raw8 &r)
{
    r.q[0][0] = lds_read_asm<OFF + 0 * JS>(addr); r.q[0][1] = lds_read_asm<OFF + 1 * JS>(addr); r.q[0][2] = lds_read_asm<OFF + 2 * JS>(addr);
    r.q[0][3] = lds_read_asm<OFF + 3 * JS>(addr); r.q[1][0] = lds_read_asm<OFF + 4 * JS>(addr); r.q[1][1] = lds_read_asm<OFF + 5 * JS>(addr);
    r.q[1][2] = lds_read_asm<OFF + 6 * JS>(addr); r.q[1][3] = lds_read_asm<OFF + 7 * JS>(addr);
}
// the same with one address per value (the x block that reaches past K: k clamped to K - 1)
template <int OFF>
__device__ __forceinline__ void read8_addr(const uint32_t *addr, Raw8 &r)
{
    r.q[0][0] = lds_read_asm<OFF>(addr[0]); r.q[0][1] = lds_read_asm<OFF>(addr[1]); r.q[0][2] = lds_read_asm<OFF>(addr[2]); r.q[0][3] = lds_read_asm<OFF>(addr[3]);
    r.q[1][0] = lds_read_asm<OFF>(addr[4]); r.q[1][1] = lds_read_asm<OFF>(addr[5]); r.q[1][2] = lds_read_asm<OFF>(addr[6]); r.q[1][3] = lds_read_asm<OFF>(addr[7]);
}
// h_{t-1} of the wave's own 16 elements of row block RB: rows (e & 3) + 8 (e >> 2) of the block (+ 4 lh in the address)
template <int RB>
__device__ __forceinline__ void read_own(uint32_t hw0, float *hv)
{
    constexpr int S = BF_HS * 4, B0 = RB * 32 * S;
    hv[0] = lds_read_asm<B0 + 0 * S>(hw0);   hv[1] = lds_read_asm<B0 + 1 * S>(hw0);   hv[2] = lds_read_asm<B0 + 2 * S>(hw0);   hv[3] = lds_read_asm<B0 + 3 * S>(hw0);
    hv[4] = lds_read_asm<B0 + 8 * S>(hw0);   hv[5] = lds_read_asm<B0 + 9 * S>(hw0);   hv[6] = lds_read_asm<B0 + 10 * S>(hw0);  hv[7] = lds_read_asm<B0 + 11 * S>(hw0);
    hv[8] = lds_read_asm<B0 + 16 * S>(hw0);  hv[9] = lds_read_asm<B0 + 17 * S>(hw0);  hv[10] = lds_read_asm<B0 + 18 * S>(hw0); hv[11] = lds_read_asm<B0 + 19 * S>(hw0);
    hv[12] = lds_read_asm<B0 + 24 * S>(hw0); hv[13] = lds_read_asm<B0 + 25 * S>(hw0); hv[14] = lds_read_asm<B0 + 26 * S>(hw0); hv[15] = lds_read_asm<B0 + 27 * S>(hw0);
}
// every outstanding LDS read has landed; ties the wait to the value registers so that no use can be scheduled above it
__device__ __forceinline__ void lds_landed2_wb(Raw8 *raw, f32x4 &wv)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0].q[0]), "+v"(raw[0].q[1]), "+v"(raw[1].q[0]), "+v"(raw[1].q[1]), "+v"(wv)::"memory");
}
__device__ __forceinline__ void wb_read1(uint32_t addr, f32x4 &wv)
{
    constexpr int S = BF_HS * 4;
    wv[0] = lds_read_asm<0 * S>(addr); wv[1] = lds_read_asm<1 * S>(addr); wv[2] = lds_read_asm<2 * S>(addr); wv[3] = lds_read_asm<3 * S>(addr);
}

template <int SPL>
__global__ __launch_bounds__(256, 2) void gru_layer_bf16_kernel(const LayerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // xS [K4][64] | hS [64][129]
    constexpr int H = BF_H, HS = BF_HS, BM = BF_BM, LPK = BM / 4, KPI = 64 / LPK;      // DMA: 16 lanes per input, four inputs per instruction
    constexpr int NP = SPL == 3 ? 6 : 3;
    constexpr int PW[6] = {0, 0, 1, 0, 2, 1}, PA[6] = {0, 1, 0, 2, 0, 1};
    const int lane = threadIdx.x & 63, chunk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_row0 = blockIdx.x * BM;
    const int li = lane & 31, lh = lane >> 5;
    const int K = a.K, K4 = (K + 3) & ~3, KBx = a.KBx, KB = KBx + BF_KBH;
    float *xS = sm, *hS = sm + K4 * BM;
    for (int i = threadIdx.x; i < BM * HS; i += 256) hS[i] = 0.f;          // h0 = 0 (gru/gru_model.py:27)

    const float *bias = a.w + (size_t)chunk * chunk_floats(a.KPx, a.KPh) + (size_t)(a.KPx + a.KPh) * 3 * 64;
    constexpr float LOG2E = 1.44269504088896341f;
    const float nb_r = -LOG2E * bias[li], nb_z = -LOG2E * bias[32 + li], nb_n = 2.0f * LOG2E * bias[64 + li], b_hn = bias[96 + li];
    const uint32_t rowB = (uint32_t)a.B * 4u;
    const rsrc_t rw = make_rsrc(a.wbf + (size_t)chunk * KB * 3 * SPL * 256, (uint32_t)KB * 3 * SPL * 1024);
    const uint32_t wl = (uint32_t)lane * 16u;

    const uint32_t dvoff = (uint32_t)(lane / LPK) * rowB + (uint32_t)(tile_row0 + 4 * (lane % LPK)) * 4u;
    auto stage_x = [&](int t) {
        const rsrc_t rx = make_rsrc(a.xs + (size_t)t * K * a.B, (uint32_t)K * rowB);      // inputs past K read as zero (range check)
        for (int j = chunk; j < K4 / KPI; j += 4)
            stage_dma16(rx, xS + j * 256, dvoff, __builtin_amdgcn_readfirstlane((uint32_t)(KPI * j) * rowB));
    };
    const uint32_t xS_b = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)xS;
    const uint32_t hS_b = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)hS;
    const uint32_t ax0 = xS_b + (uint32_t)(8 * lh * BM + li) * 4u;                 // k = 16 kb + 8 lh + j, row rb * 32 + li
    const uint32_t ah0 = hS_b + (uint32_t)(li * HS + 8 * lh) * 4u;
    const uint32_t hw0 = hS_b + (uint32_t)(4 * lh * HS + chunk * 32 + li) * 4u;

    stage_x(0);
    // h_t -> seq_out ([H][B]: consecutive trajectories of a hidden unit are contiguous) one step late: group i of a wave = hidden units chunk * 32 + 4 i + (lane >> 4), lane & 15 = four
    // consecutive trajectories: a store instruction writes four 256-byte runs
    const uint32_t wb_rd = hS_b + (uint32_t)(4 * (lane & 15) * HS + chunk * 32 + (lane >> 4)) * 4u;
    const uint32_t wb_vo = (tile_row0 + 4 * (lane & 15) < a.B) ? (uint32_t)(lane >> 4) * rowB + (uint32_t)(tile_row0 + 4 * (lane & 15)) * 4u : 0x80000000u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto req_w = [&](int kb, u32x4 (*W)[SPL]) {
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int sp = 0; sp < SPL; sp++)
                W[g][sp] = buf_load4(rw, wl, __builtin_amdgcn_readfirstlane((uint32_t)((kb * 3 + g) * SPL + sp) * 1024u));
    };
    auto req_a = [&](int kb, Raw8 *raw) {
        if (kb < KBx) {
            if (16 * kb + 16 <= K) {
                const uint32_t ad = ax0 + (uint32_t)kb * (16 * BM * 4);
                read8<0, BM * 4>(ad, raw[0]); read8<128, BM * 4>(ad, raw[1]);
            } else {
                uint32_t ad[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    int k = 16 * kb + 8 * lh + j;
                    k = k < K ? k : K - 1;
                    ad[j] = xS_b + (uint32_t)(k * BM + li) * 4u;
                }
                read8_addr<0>(ad, raw[0]); read8_addr<128>(ad, raw[1]);
            }
        } else {
            const uint32_t ad = ah0 + (uint32_t)(kb - KBx) * 64u;
            read8<0, 4>(ad, raw[0]); read8<32 * HS * 4, 4>(ad, raw[1]);
        }
    };

    u32x4 W[2][3][SPL];
    req_w(0, W[0]);
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
        const bool wbk = t > 0 && a.seq_out;
        const rsrc_t rp = make_rsrc(wbk ? a.seq_out + (size_t)(t - 1) * H * a.B : nullptr, wbk ? (uint32_t)H * rowB : 0u);

        // one k-block.  The weight request of the NEXT block is UNCONDITIONAL (the last block re-requests itself): behind
        // `if (more)` hipcc's wait-count pass loses the number of loads younger than the fragment an MFMA needs at the merge and
        // waits for vmcnt(0) -- the NEXT block's weights -- in front of every block's MFMAs (the same disease as gru_kernels.hip's
        // mfma_part, round 4).
#define OSB2_BODY(XP, CUR, kb)                                                                                                   \
        {                                                                                                                        \
            const int kn = (kb) + 1 < KB ? (kb) + 1 : (kb);                                                                      \
            const int kw = (kb) < 8 ? (kb) : 7;                                                                                  \
            f32x4 wv;                                                                                                            \
            Raw8 raw[2];                                                                                                         \
            u32x4 Af[SPL];                                                                                                       \
            wb_read1(wb_rd + (uint32_t)kw * 16u, wv);                                                                            \
            req_w(kn, W[(CUR) ^ 1]);                          /* unconditional: see below */                                     \
            req_a((kb), raw);                                                                                                    \
            lds_landed2_wb(raw, wv);                                                                                             \
            buf_store4(rp, (kb) < 8 ? wb_vo : 0x80000000u, __builtin_amdgcn_readfirstlane((uint32_t)(chunk * 32 + 4 * kw) * rowB), wv);   \
            _Pragma("unroll") for (int rb = 0; rb < 2; rb++) {                                                                    \
                split8<SPL>(raw[rb], Af);                                                                                        \
                _Pragma("unroll") for (int g = 0; g < 3; g++) {                                                                   \
                    const int G = g < 2 ? g : ((XP) ? 2 : 3);                                                                    \
                    _Pragma("unroll") for (int pi = 0; pi < NP; pi++)                                                             \
                        acc[rb][G] = mfma_bf16(Af[PA[pi]], W[CUR][g][PW[pi]], acc[rb][G]);                                       \
                }                                                                                                                \
            }                                                                                                                    \
        }
        for (int kb = 0; kb < KBx; kb += 2) {
            OSB2_BODY(true, 0, kb)
            OSB2_BODY(true, 1, kb + 1)
        }
        for (int kb = KBx; kb < KB; kb += 2) {
            OSB2_BODY(false, 0, kb)
            OSB2_BODY(false, 1, kb + 1)
        }
#undef OSB2_BODY
        OSL_TS(1)                                        // gate GEMM (both parts)
        float hv[2][16];
        read_own<0>(hw0, hv[0]); read_own<1>(hw0, hv[1]);
        lds_barrier();                                   // barrier 1: xS and hS are free
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
            asm volatile("" : "+v"(hv[rb][0]), "+v"(hv[rb][1]), "+v"(hv[rb][2]), "+v"(hv[rb][3]), "+v"(hv[rb][4]), "+v"(hv[rb][5]), "+v"(hv[rb][6]),
                         "+v"(hv[rb][7]), "+v"(hv[rb][8]), "+v"(hv[rb][9]), "+v"(hv[rb][10]), "+v"(hv[rb][11]), "+v"(hv[rb][12]), "+v"(hv[rb][13]),
                         "+v"(hv[rb][14]), "+v"(hv[rb][15]));
        OSL_TS(2)                                        // own-h reads + barrier 1
        if (t + 1 < a.T) stage_x(t + 1);
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            using osk::f2;
#pragma unroll
            for (int pr = 0; pr < 8; pr++) {
                const int e0 = 2 * pr, e1 = e0 + 1;
                const CellPair cp = gru_cell_pair((f2){acc[rb][0][e0], acc[rb][0][e1]}, (f2){acc[rb][1][e0], acc[rb][1][e1]},
                                                  (f2){acc[rb][2][e0], acc[rb][2][e1]}, (f2){acc[rb][3][e0], acc[rb][3][e1]},
                                                  (f2){hv[rb][e0], hv[rb][e1]}, nb_r, nb_z, nb_n, b_hn);
                const f2 hn = cp.hn;
                lds_write_asm<0>(hw0 + (uint32_t)((rb * 32 + (e0 & 3) + 8 * (e0 >> 2)) * HS) * 4u, hn[0]);
                lds_write_asm<0>(hw0 + (uint32_t)((rb * 32 + (e1 & 3) + 8 * (e1 >> 2)) * HS) * 4u, hn[1]);
            }
        }
        OSL_TS(3)                                        // DMA issue + cell update
        req_w(0, W[0]);                                  // the next step's first weight block: the DMA is older than these 3 SPL loads
        if (SPL == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        lds_barrier();                                   // barrier 2: h_t and x_{t+1} are in LDS
        OSL_TS(4)
    }
    {
        const rsrc_t rs = make_rsrc(a.seq_out ? a.seq_out + (size_t)(a.T - 1) * H * a.B : nullptr, a.seq_out ? (uint32_t)H * rowB : 0u);
        const rsrc_t rl = make_rsrc(a.h_last, a.h_last ? (uint32_t)H * rowB : 0u);
        for (int i = 0; i < 8; i++) {
            f32x4 wv;
            wb_read1(wb_rd + (uint32_t)i * 16u, wv);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv)::"memory");
            const uint32_t sof = __builtin_amdgcn_readfirstlane((uint32_t)(chunk * 32 + 4 * i) * rowB);
            buf_store4(rs, wb_vo, sof, wv);
            buf_store4(rl, wb_vo, sof, wv);
        }
    }
#ifdef OS_LAYER_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("gru_layer_bf16_kernel<%d> K=%d cycles per step (one wave; the SIMD's other wave belongs to the CU's second workgroup): gate GEMM %llu | own-h reads + barrier 1 %llu | DMA issue + cell %llu | DMA wait + barrier 2 %llu | sum %llu\n",
               SPL, a.K, ts_sum[1] / a.T, ts_sum[2] / a.T, ts_sum[3] / a.T, ts_sum[4] / a.T, (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4]) / a.T);
#endif
}

}  // namespace osg

using namespace osg;

// Launches gru_layer_bf16_kernel for this layer if the context opted in and the shape is the kernel's; *done says whether it did.
// The bf16 term image of the active weight slot is (re)built here, on the launch stream, from the slot's fp32 fragment image --
// the same snapshot of the caller's weights every other GRU kernel uses.
int os_gru_try_layer_bf16(os_ctx *ctx, const LayerArgs &a, hipStream_t s, bool *done)
{
    *done = false;
    const int mode = ctx->gru_split_bf16, spl = mode & 3;
    if (spl != 2 && spl != 3) return 0;
    const bool any_batch = (mode & OS_GRU_SPLIT_ANY_BATCH) != 0;
    os_ctx::GruSlot *slot = ctx->gru_slot;
    const size_t lds = ((size_t)((a.K + 3) & ~3) * BF_BM + (size_t)BF_BM * BF_HS) * sizeof(float);
    if (!slot || a.H != BF_H || a.sv_r || a.xs_btf || a.gi || a.B % 4 != 0 || lds > 80 * 1024 || (size_t)a.K * a.B * 4 >= ((size_t)1 << 31) ||
        (size_t)a.H * a.B * 4 >= ((size_t)1 << 31))
        return 0;
    if (!any_batch && (a.B + BF_BM - 1) / BF_BM < 2 * ctx->cu_count) return 0;   // fewer than two tiles per CU: the fp32 small-batch kernels
    // which layer of the loaded model is this?
    const os_gru_dims &d = ctx->gru;
    size_t off = 0, boff = 0;
    int layer = -1;
    for (int l = 0; l < d.num_layers; l++) {
        const int K = l == 0 ? d.input_size : d.hidden_size;
        if (slot->packed + off == a.w && K == a.K) { layer = l; break; }
        off += os_layer_packed_floats(K, d.hidden_size);
        boff += bf_layer_dwords(K, spl);
    }
    if (layer < 0) return 0;
    if (slot->bf_spl != spl) {
        size_t total = 0;
        for (int l = 0; l < d.num_layers; l++) total += bf_layer_dwords(l == 0 ? d.input_size : d.hidden_size, spl);
        if (os_ensure_scratch(ctx, &slot->bf, &slot->bf_cap, total)) return -10;
        size_t o = 0, bo = 0;
        for (int l = 0; l < d.num_layers; l++) {
            const int K = l == 0 ? d.input_size : d.hidden_size;
            uint32_t *dst = reinterpret_cast<uint32_t *>(slot->bf) + bo;
            if (spl == 3) hipLaunchKernelGGL(gru_pack_bf16_kernel<3>, dim3(64), dim3(256), 0, s, slot->packed + o, (K + 1) / 2, BF_H / 2, bf_kbx(K), dst);
            else hipLaunchKernelGGL(gru_pack_bf16_kernel<2>, dim3(64), dim3(256), 0, s, slot->packed + o, (K + 1) / 2, BF_H / 2, bf_kbx(K), dst);
            o += os_layer_packed_floats(K, d.hidden_size);
            bo += bf_layer_dwords(K, spl);
        }
        OS_HIP(ctx, hipGetLastError());
        slot->bf_spl = spl;
    }
    if (!ctx->bf16_layer_attr_set) {
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        OS_HIP(ctx, hipFuncSetAttribute((const void *)gru_layer_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        ctx->bf16_layer_attr_set = true;
    }
    LayerArgs b = a;
    b.wbf = reinterpret_cast<const uint32_t *>(slot->bf) + boff;
    b.KBx = bf_kbx(a.K);
    const int slot_p = os_prof_begin(ctx, OS_PHASE_GRU_LAYER, s, spl == 3 ? "gru_layer_bf16_kernel<3>" : "gru_layer_bf16_kernel<2>");
    const dim3 grid((a.B + BF_BM - 1) / BF_BM), block(256);
    if (spl == 3) hipLaunchKernelGGL(gru_layer_bf16_kernel<3>, grid, block, lds, s, b);
    else hipLaunchKernelGGL(gru_layer_bf16_kernel<2>, grid, block, lds, s, b);
    os_prof_end(ctx, slot_p, s);
    OS_HIP(ctx, hipGetLastError());
    *done = true;
    return 0;
}

extern "C" int os_gru_set_split_bf16(os_ctx *ctx, int32_t mode)
{
    OS_CHECK_CTX(ctx);
    const int spl = mode & 3;
    if ((mode & ~(3 | OS_GRU_SPLIT_ANY_BATCH | OS_GRU_SPLIT_TRAIN)) || spl == 1)
        return os_fail(ctx, -4, "os_gru_set_split_bf16: mode is 0 (exact fp32), 2 or 3 bf16 terms, optionally | OS_GRU_SPLIT_ANY_BATCH | OS_GRU_SPLIT_TRAIN");
    ctx->gru_split_bf16 = mode;
    return 0;
}
