// vit_kernels.hip -- encoder half of the reference's ViT auto-encoder (transformer/transformer_model.py:113-135; optional
// row A10 / BASELINE config 5): depth frames (N,1,224,224) -> 128-d latent in (0,1) that is appended to the 60 Kalman
// features (gru/gru_test.py:119-136).  The blocks are timm-0.3.2 `Block`s (pre-LN, fused qkv with bias, GELU(erf) MLP,
// LayerNorm eps 1e-5, no drop-path); timm itself is absent from the build image, so this follows the published
// definition of that release -- parity UNPINNED, checked only against this repo's own float64
// restatement (oracle/vit_oracle.py).
//
// Everything on the path is a hand-written gfx950 kernel: the dense projections ([N*197 x 128] x [128 x 384/128/512],
// [.. x 512] x [512 x 128], patch embedding [N*196 x 256] x [256 x 128]) run on ONE fp32-MFMA GEMM kernel with fused
// prologue / epilogues (vit_gemm_kernel: patch extraction while staging, bias, bias + GELU, bias + residual, patch bias +
// position table); LayerNorm, per-(image, head) attention on the matrix cores, and the final LayerNorm + sigmoid of the cls
// token are separate kernels.  No vendor BLAS.
#include "launch.hpp"

#include <math.h>

namespace osv {

struct VitDims { int img, patch, dim, depth, heads, mlp; };

__host__ __device__ inline int grid_of(const VitDims &d) { return d.img / d.patch; }
__host__ __device__ inline int ntok(const VitDims &d) { return grid_of(d) * grid_of(d) + 1; }

// LayerNorm over rows of D (D <= 256, one wave per row, eps 1e-5)
__global__ void layernorm_kernel(size_t M, int D, const float *X, const float *w, const float *b, float *Y, size_t x_stride)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *x = X + row * x_stride;
    float v[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; v[j] = c < D ? x[c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; const float t = c < D ? v[j] - mean : 0.f; q += t * t; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; if (c < D) Y[row * D + c] = (v[j] - mean) * rstd * w[c] + b[c]; }
}

// Attention for one (image, head): QKV rows [L][3D] (bias already added by the projection's epilogue), head h owns columns h*hd..; K and V of the head staged in LDS,
// one query per lane with an online softmax; output O[row][h*hd + :].  hd <= 64.
template <int HD>
__global__ __launch_bounds__(256) void attention_kernel(int L, int D, int heads, const float *QKV, const float *qkv_b, float *O)
{
    extern __shared__ float sm[];            // K [L][HD+1], V [L][HD+1]
    float *Ks = sm, *Vs = sm + (size_t)L * (HD + 1);
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *base = QKV + (size_t)n * L * 3 * D;
    for (int i = threadIdx.x; i < L * HD; i += blockDim.x) {
        const int l = i / HD, c = i % HD;
        Ks[l * (HD + 1) + c] = base[(size_t)l * 3 * D + D + h * HD + c];
        Vs[l * (HD + 1) + c] = base[(size_t)l * 3 * D + 2 * D + h * HD + c];
    }
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    for (int qi = threadIdx.x; qi < L; qi += blockDim.x) {
        float q[HD], acc[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) { q[c] = base[(size_t)qi * 3 * D + h * HD + c] * scale; acc[c] = 0.f; }
        float mx = -3.0e38f, den = 0.f;
        for (int l = 0; l < L; l++) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; c++) s += q[c] * Ks[l * (HD + 1) + c];
            const float mn = fmaxf(mx, s);
            const float corr = __expf(mx - mn), p = __expf(s - mn);
            den = den * corr + p;
#pragma unroll
            for (int c = 0; c < HD; c++) acc[c] = acc[c] * corr + p * Vs[l * (HD + 1) + c];
            mx = mn;
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int c = 0; c < HD; c++) O[((size_t)n * L + qi) * D + h * HD + c] = acc[c] * inv;
    }
}

// The same attention on the matrix cores for head_dim = 32 (the reference's ViT: 4 heads x 32).  One workgroup per
// (image, head), K and V (+bias) of the head in LDS; a wave owns a 32-query tile.  Per tile: S^T = K Q^T as seven 32x32
// v_mfma_f32_32x32x2_f32 tiles (exact fp32), keys beyond L masked, the softmax statistics of a lane's query in-lane plus one
// exchange between the wave halves, O^T = V^T P^T with P^T taken straight from the accumulator registers, scaled by 1 / sum
// (details at the tile loop).
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int ATT_HD = 32, ATT_KS = ATT_HD + 1, ATT_MAXT = 8;      // up to 256 tokens

// Reduction over the 32 lanes of a wave half (the columns of a C-layout row) on the VALU's data-parallel-primitive paths:
// two quad permutes, row_half_mirror, row_mirror (all fused into the max / add as DPP operands), then v_permlane16_swap
// pairs the two 16-lane rows of the half.  The __shfl_xor form was five ds_bpermute round trips per reduction, 160 per
// query tile.  (Inline asm for the swap: hipcc of ROCm 7.2 drops the second result of __builtin_amdgcn_permlane16_swap.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void rows_swap16(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// lanes 32-63 of a <-> lanes 0-31 of b (a = b = v on entry: afterwards a holds the low half's value in both halves, b the high
// half's); inline asm for the same reason as rows_swap16
__device__ __forceinline__ void halves_swap32(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
    float a = v, b = v;
    rows_swap16(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_sum(float v)
{
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    float a = v, b = v;
    rows_swap16(a, b);
    return a + b;
}

// NW waves per workgroup: with eight, the seven query tiles of the reference's 197 tokens run side by side (one per wave, two
// waves per SIMD covering each other's softmax and LDS round trips) instead of two rounds over four waves with one idle in
// the second, and the head's K / V are still staged once.
template <int NT /* key / query tiles = ceil(L / 32) */, int NW = 8>
__global__ __launch_bounds__(NW * 64, 1) void attention_mfma_kernel(int L, int D, int heads, const float *QKV, const float *qkv_b, float *O)
{
    extern __shared__ float sm[];            // K [L][33] | V [L][33]
    float *Ks = sm, *Vs = sm + (size_t)L * ATT_KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *base = QKV + (size_t)n * L * 3 * D;
    // staging with float4 loads, four in flight per thread (a scalar loop here was one exposed HBM round trip per element:
    // it dominated the kernel)
    {
        const int c4 = (threadIdx.x & 7) * 4;                       // 8 float4 per 32-wide head row
#pragma unroll 4
        for (int l = threadIdx.x >> 3; l < L; l += NW * 8) {
            const float4 kv = *reinterpret_cast<const float4 *>(base + (size_t)l * 3 * D + D + h * ATT_HD + c4);
            const float4 vv = *reinterpret_cast<const float4 *>(base + (size_t)l * 3 * D + 2 * D + h * ATT_HD + c4);
            float *kd = Ks + l * ATT_KS + c4, *vd = Vs + l * ATT_KS + c4;
            kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
            vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
        }
    }
    __syncthreads();
    // 1/sqrt(d) and log2(e) folded into Q: the softmax numerators are exp2(S' - max'), one v_exp_f32 per score and no multiply
    const float scale = rsqrtf((float)ATT_HD) * 1.44269504088896341f;
    for (int qt = wave; qt < NT; qt += NW) {
        const int r0 = 32 * qt;
        // Q tile as A fragments: lane -> (row r0 + li, dim 2q + lh)
        const int qrow = r0 + li < L ? r0 + li : L - 1;
        float qa[ATT_HD / 2];
#pragma unroll
        for (int q = 0; q < ATT_HD / 2; q++)
            qa[q] = base[(size_t)qrow * 3 * D + h * ATT_HD + 2 * q + lh] * scale;
        // TRANSPOSED score tiles S^T = K Q^T (A = K rows, B = Q^T): in the C layout a lane then owns ONE query (column li) and
        // its registers run over the keys, (e & 3) + 8 (e >> 2) + 4 lh of each 32-key tile.  The softmax statistics of a query
        // are in-lane reductions over 7 x 16 registers plus one exchange between the two wave halves (v_permlane32_swap) --
        // not sixteen cross-lane reductions per tile -- and P^T never leaves the registers: O^T = V^T P^T takes it as its B
        // operand directly, with the reduction's k-pairs renumbered to the C layout's key order (pair e = keys a_e, a_e + 4
        // with a_e = (e & 3) + 8 (e >> 2); a sum does not care about the order) and the V^T fragments fetched in that order.
        // (The P tile used to go through LDS, 16 writes + 16 reads per key tile, and the row statistics were ~290 VALU
        // instructions per query tile on the pipe the MFMAs need.)  The 16 K fragments of tile jt+1 are requested from LDS
        // before the 16 MFMAs of tile jt issue.
        f32x16v S[NT];
        float kb[2][ATT_HD / 2];
        {
            const int key = li < L ? li : L - 1;
#pragma unroll
            for (int q = 0; q < ATT_HD / 2; q++) kb[0][q] = Ks[key * ATT_KS + 2 * q + lh];
        }
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            if (jt + 1 < NT) {
                const int key = 32 * (jt + 1) + li < L ? 32 * (jt + 1) + li : L - 1;
#pragma unroll
                for (int q = 0; q < ATT_HD / 2; q++) kb[(jt + 1) & 1][q] = Ks[key * ATT_KS + 2 * q + lh];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) S[jt][e] = 0.f;
#pragma unroll
            for (int q = 0; q < ATT_HD / 2; q++) S[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[jt & 1][q], qa[q], S[jt], 0, 0, 0);
            if (32 * jt + 32 > L) {                      // tile with keys past L: mask them (register index = key)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (32 * jt + (e & 3) + 8 * (e >> 2) + 4 * lh >= L) S[jt][e] = -3.0e38f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // softmax statistics of this lane's query: in-lane over the keys it holds, then the other wave half's share
        float m = S[0][0];
#pragma unroll
        for (int jt = 0; jt < NT; jt++)
#pragma unroll
            for (int e = 0; e < 16; e++) m = fmaxf(m, S[jt][e]);
        {
            float ma = m, mb = m;
            halves_swap32(ma, mb);
            m = fmaxf(ma, mb);
        }
        float den = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; jt++)
#pragma unroll
            for (int e = 0; e < 16; e++) { const float pv = __builtin_amdgcn_exp2f(S[jt][e] - m); S[jt][e] = pv; den += pv; }
        {
            float da = den, db = den;
            halves_swap32(da, db);
            den = da + db;
        }
        // O^T = V^T P^T: A fragment (dim li, keys a_e + 4 lh of the tile) from LDS, B fragment = S[jt][e]
        f32x16v acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        float vbf[2][16];
        {
#pragma unroll
            for (int e = 0; e < 16; e++) { const int key = (e & 3) + 8 * (e >> 2) + 4 * lh; vbf[0][e] = Vs[(key < L ? key : L - 1) * ATT_KS + li]; }
        }
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            if (jt + 1 < NT) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = 32 * (jt + 1) + (e & 3) + 8 * (e >> 2) + 4 * lh;                            // P is 0 past L
                    vbf[(jt + 1) & 1][e] = Vs[(key < L ? key : L - 1) * ATT_KS + li];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vbf[jt & 1][e], S[jt][e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane = query r0 + li, registers = head dims (e & 3) + 8 (e >> 2) + 4 lh: four 16-byte stores per lane
        if (r0 + li < L) {
            const float inv = 1.0f / den;
            float *op = O + ((size_t)n * L + r0 + li) * D + h * ATT_HD + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(op + 8 * g) = make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
        }
    }
}

__device__ __forceinline__ void vit_lds_dma16(const float *g, float *l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
// barrier that waits for this wave's LDS traffic only (a __syncthreads() also waits for the output stores just issued)
__device__ __forceinline__ void vit_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The same attention as a PERSISTENT workgroup: K and V of the NEXT (image, head) arrive by LDS-DMA (global_load_lds, no staging
// registers) into the other half of a double buffer while the current one is being consumed.  In attention_mfma_kernel the
// staging of a head's K / V was 7.2 k of a workgroup's ~60 k cycles with everything else idle (one workgroup per CU: 256
// registers per lane); prefetching into registers instead spilled (131).  The DMA writes whole 16-byte pieces at consecutive
// LDS addresses, so K's rows are NINE pieces long with the ninth lane of every nine switched off: a pitch of 36 floats makes
// the fragment read of 32 keys at one column 2-way conflicted (32 floats: 32-way; an XOR swizzle: 4-way and eight more
// address registers).  V is read along its rows: pitch 32.
// K / V of one (image, head) -> dst by LDS-DMA (a real call: inlined, its address arithmetic shares the register allocation of
// the attention body, which has none to spare)
template <int NW, int KP>
__device__ __attribute__((noinline)) void attention_stage_dma(const float *QKV, int L, int D, int heads, int item, float *dst, int wave_)
{
    // (no threadIdx here: a callee that reads it makes the caller keep the packed work-item ids alive for v31 across its whole
    // body -- one more spilled register there; the lane id is the exec-mask count, the wave comes as an argument)
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(wave_);
    const int npk = L * (KP / 4), npieces = npk + L * 8;
    const int n = item / heads, h = item % heads;
    const float *base = QKV + (size_t)n * L * 3 * D + h * ATT_HD;
    for (int g0 = wave * 64; g0 < npieces; g0 += NW * 64) {               // wave-uniform: one DMA instruction = 64 consecutive piece slots
        const int g = g0 + lane;
        if (g < npieces) {
            const bool isv = g >= npk;
            const int l = isv ? (g - npk) >> 3 : g / (KP / 4), pc = isv ? (g - npk) & 7 : g % (KP / 4);
            const float *src = base + (size_t)l * 3 * D + (isv ? 2 * D : D) + 4 * pc;
            if (pc < 8) vit_lds_dma16(src, dst + (size_t)g0 * 4);          // the ninth slot of a K row is padding
        }
    }
}

template <int NT, int NW = 8>
__global__ __launch_bounds__(NW * 64, 1) void attention_mfma_dma_kernel(int L, int D, int heads, int nitems, const float *QKV, float *O)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];            // 2 x (K [L][36] | V [L][32])
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    constexpr int KP = 36;                                                // K row pitch in floats
    const int bufw = L * (KP + 32);                                       // floats per buffer
    int item = blockIdx.x;
    if (item < nitems) attention_stage_dma<NW, KP>(QKV, L, D, heads, item, sm, wave);
    bool stored = false;                                                  // this wave issued output stores in the previous item
    // This wave's Q tile (NT <= NW: one tile per wave and head), RAW: the fragments of the NEXT head are requested right after
    // the score MFMAs of the current one (their registers are dead from there on) and scaled when that head starts -- the
    // sixteen strided loads used to sit exposed at the top of every tile (2-5 k of its ~50 k cycles).
    static_assert(NT <= NW, "one query tile per wave");
    float qa[ATT_HD / 2];
    auto loadq = [&](int it_, int li__, int lh__) {
        const int n_ = it_ / heads, h_ = it_ % heads;
        const int qrow = 32 * wave + li__ < L ? 32 * wave + li__ : L - 1;
        const float *qp = QKV + (size_t)n_ * L * 3 * D + (size_t)qrow * 3 * D + h_ * ATT_HD + lh__;
#pragma unroll
        for (int q = 0; q < ATT_HD / 2; q++) qa[q] = qp[2 * q];
    };
    if (item < nitems && wave < NT) loadq(item, li, lh);
    for (int it = 0; item < nitems; item += gridDim.x, it++) {
        // my DMA pieces of this item have landed: they are older than the (at most four) output stores of the previous item,
        // which need not be waited for (loads and stores retire in issue order)
        if (stored) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f70);
        vit_lds_barrier();
        float *Kb = sm + (it & 1) * bufw, *Vb = Kb + L * KP;
        if (item + (int)gridDim.x < nitems) attention_stage_dma<NW, KP>(QKV, L, D, heads, item + gridDim.x, sm + ((it + 1) & 1) * bufw, wave);
        const int n = item / heads, h = item % heads;
        const float *base = QKV + (size_t)n * L * 3 * D;
        stored = false;
        // opaque per-iteration copies: everything below that depends only on (L, lane) -- 112 clamped V row addresses, the key
        // masks -- is invariant across the item loop, and hoisted out of it it stays live across the whole body (60 spills)
        // (the lane id itself is re-derived per item from the exec mask count: copies of li / lh would keep those two registers and
        // the row addresses built from them alive -- and spilled, 8 registers -- across the body)
        int L_ = L, lane_;
        asm volatile("" : "+s"(L_));
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int li_ = lane_ & 31, lh_ = lane_ >> 5;
        // 1/sqrt(d) and log2(e) folded into Q: the softmax numerators are exp2(S' - max'), one v_exp_f32 per score and no multiply
        const float scale = rsqrtf((float)ATT_HD) * 1.44269504088896341f;
        for (int qt = wave; qt < NT; qt += NW) {
            const int r0 = 32 * qt;
            // Q tile as A fragments: lane -> (row r0 + li_, dim 2q + lh_), prefetched raw (see loadq)
#pragma unroll
            for (int q = 0; q < ATT_HD / 2; q++) qa[q] *= scale;
            // TRANSPOSED score tiles S^T = K Q^T (A = K rows, B = Q^T): in the C layout a lane then owns ONE query (column li_) and
            // its registers run over the keys, (e & 3) + 8 (e >> 2) + 4 lh_ of each 32-key tile.  The softmax statistics of a query
            // are in-lane reductions over 7 x 16 registers plus one exchange between the two wave halves (v_permlane32_swap) --
            // not sixteen cross-lane reductions per tile -- and P^T never leaves the registers: O^T = V^T P^T takes it as its B
            // operand directly, with the reduction's k-pairs renumbered to the C layout's key order (pair e = keys a_e, a_e + 4
            // with a_e = (e & 3) + 8 (e >> 2); a sum does not care about the order) and the V^T fragments fetched in that order.
            // (The P tile used to go through LDS, 16 writes + 16 reads per key tile, and the row statistics were ~290 VALU
            // instructions per query tile on the pipe the MFMAs need.)  The 16 K fragments of tile jt+1 are requested from LDS
            // before the 16 MFMAs of tile jt issue.
            f32x16v S[NT];
            float kb[2][ATT_HD / 2];
            {
                const int key = li_ < L_ ? li_ : L_ - 1;
#pragma unroll
                for (int q = 0; q < ATT_HD / 2; q++) kb[0][q] = Kb[key * KP + 2 * q + lh_];
            }
#pragma unroll
            for (int jt = 0; jt < NT; jt++) {
                if (jt + 1 < NT) {
                    const int key = 32 * (jt + 1) + li_ < L_ ? 32 * (jt + 1) + li_ : L_ - 1;
#pragma unroll
                    for (int q = 0; q < ATT_HD / 2; q++) kb[(jt + 1) & 1][q] = Kb[key * KP + 2 * q + lh_];
                }
#pragma unroll
                for (int e = 0; e < 16; e++) S[jt][e] = 0.f;
#pragma unroll
                for (int q = 0; q < ATT_HD / 2; q++) S[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[jt & 1][q], qa[q], S[jt], 0, 0, 0);
                if (32 * jt + 32 > L_) {                      // tile with keys past L_: mask them (register index = key)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        if (32 * jt + (e & 3) + 8 * (e >> 2) + 4 * lh_ >= L_) S[jt][e] = -3.0e38f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // softmax statistics of this lane's query: in-lane over the keys it holds, then the other wave half's share
            float m = S[0][0];
#pragma unroll
            for (int jt = 0; jt < NT; jt++)
#pragma unroll
                for (int e = 0; e < 16; e++) m = fmaxf(m, S[jt][e]);
            {
                float ma = m, mb = m;
                halves_swap32(ma, mb);
                m = fmaxf(ma, mb);
            }
            // every score MFMA has delivered (the maximum consumed their results): the Q registers are free for the next head
            __builtin_amdgcn_sched_barrier(0);
            if (item + (int)gridDim.x < nitems) loadq(item + gridDim.x, li_, lh_);
            __builtin_amdgcn_sched_barrier(0);
            float den = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; jt++)
#pragma unroll
                for (int e = 0; e < 16; e++) { const float pv = __builtin_amdgcn_exp2f(S[jt][e] - m); S[jt][e] = pv; den += pv; }
            {
                float da = den, db = den;
                halves_swap32(da, db);
                den = da + db;
            }
            // O^T = V^T P^T: A fragment (dim li_, keys a_e + 4 lh_ of the tile) from LDS, B fragment = S[jt][e]
            f32x16v acc;
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            float vbf[2][16];
            {
#pragma unroll
                for (int e = 0; e < 16; e++) { const int key = (e & 3) + 8 * (e >> 2) + 4 * lh_; vbf[0][e] = Vb[(key < L_ ? key : L_ - 1) * 32 + li_]; }
            }
#pragma unroll
            for (int jt = 0; jt < NT; jt++) {
                if (jt + 1 < NT) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int key = 32 * (jt + 1) + (e & 3) + 8 * (e >> 2) + 4 * lh_;                            // P is 0 past L_
                        vbf[(jt + 1) & 1][e] = Vb[(key < L_ ? key : L_ - 1) * 32 + li_];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vbf[jt & 1][e], S[jt][e], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane = query r0 + li_, registers = head dims (e & 3) + 8 (e >> 2) + 4 lh_: four 16-byte stores per lane
            stored = true;
            if (r0 + li_ < L_) {
                const float inv = 1.0f / den;
                float *op = O + ((size_t)n * L_ + r0 + li_) * D + h * ATT_HD + 4 * lh_;
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *reinterpret_cast<float4 *>(op + 8 * g) = make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
            }
        }

    }
}

// ---- vit_gemm_kernel: C[M x N] = A[M x K] . W[N x K]^T on v_mfma_f32_32x32x2_f32 (exact fp32) with fused ends -----------------
// Workgroup = 4 waves = a 128-row x 128-column tile; wave w owns column chunk w (32 columns) for all four 32-row blocks
// (64 accumulator registers).  A is staged 128 columns at a time (K = 128: once; 256 / 512: two / four chunks, accumulators
// kept) by LDS-DMA with the 16-byte slots of a row XOR-swizzled by (row & 7): the fragment read of 32 rows at one column is
// 4-way instead of 32-way conflicted (the DMA destination must stay linear, so the swizzle is applied to the SOURCE
// address and again on the read).  W is packed once per os_vit_load into B-fragment order (column chunk, k-pair, lane) and
// streamed from L2 by coalesced buffer loads with an 8-deep software pipeline -- the layer kernel's scheme.  Two
// workgroups per CU (64 KB LDS, <= 128 VGPRs) cover each other's staging.
// SRC: 0 row-major A [M][lda]; 1 patches taken straight from the images (row = (frame, gy, gx), column = 16 py + px).
// EPI: 0 C = acc + bias; 1 C = gelu(acc + bias) (exact erf form); 2 X += acc + bias (residual, in place);
//      3 X[frame*L + 1 + p] = acc + bias + pos[1 + p] (patch embedding into the token matrix).
struct GemmArgs {
    size_t M;
    int N, K, lda;
    const float *A;            // SRC 0
    const float *img;          // SRC 1: [frames][img][img]
    int img_size, grid, patch; // SRC 1
    const float *Wp;           // packed weights
    const float *bias;         // [N]
    float *C;                  // EPI 0/1: [M][N]; EPI 2/3: the token matrix X
    const float *pos;          // EPI 3: [L][N]
    int L;                     // EPI 3
};

constexpr int GM_BM = 128, GM_KC = 128;


__global__ void vit_pack_w_kernel(int N, int K, const float *W, float *dst)
{
    // dst[(c * K/2 + q) * 64 + lane] = W[c*32 + (lane & 31)][2q + (lane >> 5)]
    const size_t total = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = i & 63;
        const size_t cq = i >> 6;
        const int q = cq % (K / 2), c = cq / (K / 2);
        dst[i] = W[(size_t)(c * 32 + (lane & 31)) * K + 2 * q + (lane >> 5)];
    }
}

// GELU in its exact erf form (nn.GELU default of that release) with erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, on
// the hardware exp2 / rcp): ~15 VALU instructions instead of the ~40 of ocml's erff, which in a GEMM epilogue at two waves
// per SIMD cost as much as the tile's MFMAs.
__device__ __forceinline__ float gelu_erf(float v)
{
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erfz = fmaf(-poly, __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z), 1.0f);
    return 0.5f * v * (1.0f + copysignf(erfz, v));
}

// the same on a pair of values: the polynomial and the affine steps as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32), only the
// reciprocal and the exponential per component -- the tail kernel's GELU is ~10 % of its time on the pipe its MFMAs need
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v gelu_erf2(f2v v)
{
    const f2v one = {1.0f, 1.0f};
    const f2v z = __builtin_elementwise_abs(v) * 0.70710678118654752f;
    const f2v den = z * 0.3275911f + one;
    const f2v t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f2v poly = t * 1.061405429f + (f2v){-1.453152027f, -1.453152027f};
    poly = poly * t + (f2v){1.421413741f, 1.421413741f};
    poly = poly * t + (f2v){-0.284496736f, -0.284496736f};
    poly = poly * t + (f2v){0.254829592f, 0.254829592f};
    poly = poly * t;
    const f2v zz = z * z * -1.44269504088896341f;
    const f2v ex = {__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)};
    const f2v erfz = one - poly * ex;
    const f2v sg = {copysignf(erfz.x, v.x), copysignf(erfz.y, v.y)};
    return v * 0.5f * (one + sg);
}

template <int SRC, int EPI>
__global__ __launch_bounds__(256, 2) void vit_gemm_kernel(const GemmArgs a)
{
    __shared__ __attribute__((aligned(16))) float As[GM_BM * GM_KC];          // 64 KB, rows of 32 swizzled 16-byte slots
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    const size_t row0 = (size_t)blockIdx.y * GM_BM;
    const int chunk = blockIdx.x * 4 + wave;                                  // 32-column chunk of this wave
    const bool cok = chunk * 32 < a.N;
    const int KP = a.K / 2;
    const osk::rsrc_t rw = osk::make_rsrc(a.Wp + (size_t)chunk * KP * 64, cok ? (uint32_t)KP * 256u : 0u);
    const uint32_t wl = (uint32_t)lane * 4u;
    // this thread's 16 staging pieces of a chunk: piece p = threadIdx.x + 256 i -> LDS row p / 32, slot p % 32, which receives
    // the source slot (p % 32) ^ (row & 7)
    f32x16v acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[rb][e] = 0.f;
    for (int kc = 0; kc < a.K; kc += GM_KC) {
        if (kc) __syncthreads();                                              // everyone is done reading the previous chunk
        {
            // piece p = threadIdx.x + 256 i sits in LDS row (threadIdx.x >> 5) + 8 i: the swizzled slot and the in-row offset are
            // the same for all 16 pieces of a thread, the row advances by a wave-uniform stride
            const int r_lo = threadIdx.x >> 5, slot = (threadIdx.x & 31) ^ (r_lo & 7), k = kc + 4 * slot;
            // SRC 1: (frame, gy, gx) of the thread's first row by division ONCE; the other fifteen pieces are eight rows apart each,
            // i.e. gx + 8 with at most one carry into gy and n when the grid is at least eight wide (three runtime divisions per
            // piece were ~40 VALU instructions each, ~1,300 per thread and tile on the pipe the MFMAs share)
            const int G = SRC == 1 ? a.grid : 1, P = SRC == 1 ? a.patch : 1;
            uint32_t gx = 0, gy = 0, fn = 0;
            const bool carry8 = G >= 8;
            if (SRC == 1) {
                const uint32_t m0 = (uint32_t)(row0 + r_lo);
                gx = m0 % G; gy = (m0 / G) % G; fn = m0 / (G * G);
            }
            const int kpy = SRC == 1 ? k / P : 0, kpx = SRC == 1 ? k % P : 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                size_t row = row0 + r_lo + 8 * i;
                const float *src;
                if (SRC == 0) {
                    const float *rowbase = a.A + (row0 + 8 * i) * (size_t)a.lda;        // wave-uniform
                    src = rowbase + (size_t)r_lo * a.lda + k;
                    if (row >= a.M) src = a.A + (a.M - 1) * (size_t)a.lda + k;
                } else {
                    uint32_t cx = gx, cy = gy, cn = fn;
                    if (row >= a.M || !carry8) {                                  // tail rows / narrow grids: by division
                        if (row >= a.M) row = a.M - 1;
                        const uint32_t m = (uint32_t)row;
                        cx = m % G; cy = (m / G) % G; cn = m / (G * G);
                    }
                    src = a.img + ((size_t)cn * a.img_size + cy * P + kpy) * a.img_size + cx * P + kpx;   // column k = P py + px of the patch
                    gx += 8;
                    if (gx >= (uint32_t)G) { gx -= G; gy++; if (gy >= (uint32_t)G) { gy = 0; fn++; } }
                }
                vit_lds_dma16(src, &As[(wave * 64 + 256 * i) * 4]);
            }
        }
        __syncthreads();                                                      // vmcnt(0) + barrier: the chunk has landed
        // Fully unrolled over the chunk's 64 k-pairs with an 8-deep ring: W fragments from L2 (descriptor + SGPR chunk offset +
        // immediate), A fragments from LDS.  Element (row r, column c) of the chunk sits in 16-byte slot (c >> 2) ^ (r & 7) of row
        // r; for k-pair q this lane needs column 2q + lh of rows rb*32 + li: slot (q >> 1) ^ (li & 7).  Only the low three slot bits
        // depend on the lane, so eight per-lane base addresses (one per value of (q >> 1) & 7) plus immediates cover every
        // fragment read of the kernel: no address arithmetic inside the MFMA stream.
        constexpr int D8 = 8;
        float wbf[D8], abf[D8][4];
        const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(kc / 2) * 256u);
        const float *lo8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) lo8[j] = As + li * GM_KC + ((j ^ (li & 7)) << 2) + lh;
#define VIT_AFRAG(q, rb) lo8[((q) >> 1) & 7][(rb) * 32 * GM_KC + (((q) >> 1) >> 3) * 32 + ((q) & 1) * 2]
#pragma unroll
        for (int d = 0; d < D8; d++) {
            wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, wbase);
#pragma unroll
            for (int rb = 0; rb < 4; rb++) abf[d][rb] = VIT_AFRAG(d, rb);
        }
#pragma unroll
        for (int q = 0; q < GM_KC / 2; q++) {
            const int d = q & (D8 - 1);
#pragma unroll
            for (int rb = 0; rb < 4; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], acc[rb], 0, 0, 0);
            if (q + D8 < GM_KC / 2) {
                wbf[d] = osk::buf_load(rw, wl + (uint32_t)(q + D8) * 256u, wbase);
#pragma unroll
                for (int rb = 0; rb < 4; rb++) abf[d][rb] = VIT_AFRAG(q + D8, rb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef VIT_AFRAG
    }
    if (!cok) return;
    // Epilogue through a buffer descriptor over the output: offset = per-lane constant + SGPR row offset, rows past M fall
    // outside the descriptor's range and are dropped by the hardware (no per-element address arithmetic or bounds test).
    const int col = chunk * 32 + li;
    const float bv = a.bias[col];
    if (EPI == 3) {
        // (frame, patch) of the lane's first row by division once; its other 63 rows lie at most 127 further on, so with at
        // least 128 patches per frame there is at most one carry (a division per element was ~30 instructions x 64)
        const uint32_t np = (uint32_t)(a.L - 1);
        const uint32_t m0 = (uint32_t)(row0 + 4 * lh), n0 = m0 / np, p0 = m0 % np;
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t off = (uint32_t)(rb * 32 + (e & 3) + 8 * (e >> 2));
                const size_t row = row0 + 4 * lh + off;
                if (row >= a.M) continue;
                uint32_t n = n0, pch = p0 + off;
                if (np >= 128u) { if (pch >= np) { pch -= np; n++; } }
                else { n = (uint32_t)row / np; pch = (uint32_t)row % np; }
                a.C[((size_t)n * a.L + 1 + pch) * a.N + col] = acc[rb][e] + bv + a.pos[(size_t)(1 + pch) * a.N + col];
            }
        return;
    }
    const osk::rsrc_t rc = osk::make_rsrc(a.C, (uint32_t)(a.M * (size_t)a.N * 4));
    const uint32_t vo = (uint32_t)((4 * lh) * a.N + col) * 4u, rowb = (uint32_t)a.N * 4u;
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(row0 + rb * 32 + (e & 3) + 8 * (e >> 2)) * rowb);
            float v = acc[rb][e] + bv;
            if (EPI == 1) v = gelu_erf(v);
            if (EPI == 2) v += osk::buf_load(rc, vo, so);
            osk::buf_store(rc, vo, so, v);
        }
}

// ---- vit_mlp_kernel: x += fc2(gelu(fc1(y))) for a 128-row tile without the hidden activations ever leaving the CU --------
// fc1 and fc2 as two vit_gemm launches wrote and re-read the [M][mlp] hidden matrix: 2 x 413 MB per block at 1024 frames,
// as much time as the 52.8 GFLOP of the two products.  Here the hidden dimension is walked in slices of 128: a slice of
// gelu(y W1^T + b1) is formed (fc1, K = 128) into an LDS tile in the same swizzled A-operand layout as the input tile, and
// immediately consumed as a K-slice of fc2, whose 128 x 128 accumulators stay in registers across the slices.  Eight waves:
// wave = (32-column chunk w & 3, row-block pair w >> 2); per slice 128 + 128 MFMAs per wave, two barriers.
// embed_dim = 128 only (one K-chunk of input, four output chunks); LDS: input tile 64 KB + hidden slice 64 KB.
struct MlpArgs {
    size_t M;
    int Mh;                     // hidden width, a multiple of 128
    const float *Y;             // [M][128] input: already normalised (lnw == null), the residual stream itself (LayerNorm applied here),
                                // or -- with Wpp -- the attention output that the projection below turns into the residual update
    const float *lnw, *lnb;     // LayerNorm weight / bias [128], or null
    const float *Wpp, *bp;      // attention output projection (packed N = 128, K = 128; bias [128]), or null: x += y Wp^T + bp is
                                // done HERE first (needs lnw), so that everything between the attention and the next block's qkv
                                // is one pass over the rows (the projection as its own GEMM launch was three passes over X)
    float *X;                   // [M][128] residual stream, updated in place
    const float *lnw_next, *lnb_next;   // the NEXT block's first LayerNorm, or null: its output rows are written to Ynext from
    float *Ynext;                       // this kernel's epilogue (the row is complete here; no LayerNorm launch, no re-read of X)
    const float *Wqp, *bq;              // with them: the next block's qkv projection (packed N = 384, K = 128; bias [384]) of the
    float *QKVnext;                     // normalised tile straight into QKVnext [M][384] instead of Ynext (no qkv launch either)
    const float *W1p, *b1;      // fc1 packed (vit_pack_w_kernel: N = Mh, K = 128), bias [Mh]
    const float *W2p, *b2;      // fc2 packed (N = 128, K = Mh), bias [128]
    // Workgroups below nfull own 128-row tiles; the rest own `small_rows`-row tiles (32 or 64) behind them.  1024 frames are
    // 1576 full tiles = 6.16 rounds over 256 CUs, i.e. a seventh round with 40 CUs busy; as 160 quarter tiles that round costs
    // about a third of a full one (a wave skips the row blocks its tile does not have).
    int nfull, small_rows;
};

// NRB = 32-row blocks this wave owns in its tile (2 in a 128-row tile; 1 or 0 in the short tiles of the last round): a template
// parameter, not a run-time bound -- with `if (rb < nrb)` around the MFMAs of the unrolled loops the kernel ran at half speed
// (a scalar branch in front of every matrix instruction).  Every instantiation executes the same sequence of barriers.
// BM = rows per tile = 4 x the workgroup's threads: 128 (eight waves, one workgroup per CU) or 64 (four waves, two workgroups per
// CU, whose staging, LayerNorm and GELU phases then overlap each other's MFMA phases).
template <int NRB, int BM>
__device__ __forceinline__ void vit_mlp_tile(const MlpArgs &a, const size_t row0, const int nrows)
{
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) float msm[];
    constexpr int NTHR = BM * 4;
    float *As = msm, *Hs = msm + BM * D;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, lh = lane >> 5;
    const int chunk = wave & 3, rbp = wave >> 2;
    {
        // input tile: piece p = threadIdx.x + 512 i -> LDS row (threadIdx.x >> 5) + 16 i, slot p % 32 receives source slot
        // (p % 32) ^ (row & 7) (the row's low three bits do not change with i)
        const int r_lo = threadIdx.x >> 5, slot = (threadIdx.x & 31) ^ (r_lo & 7);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if ((NTHR / 32) * i >= nrows) break;                          // wave-uniform
            size_t row = row0 + r_lo + (NTHR / 32) * i;
            if (row >= a.M) row = a.M - 1;
            vit_lds_dma16(a.Y + row * D + 4 * slot, &As[(wave * 64 + NTHR * i) * 4]);
        }
    }
    f32x16v acc2[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc2[rb][e] = 0.f;
    const uint32_t wl = (uint32_t)lane * 4u;
    // eight per-lane base addresses cover every A-fragment read of a swizzled tile (see vit_gemm_kernel)
    int lo8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) lo8[j] = (rbp * 64 + li) * D + ((j ^ (li & 7)) << 2) + lh;
#define MLP_AFRAG(T, q, rb) (T)[lo8[((q) >> 1) & 7] + (rb) * 32 * D + (((q) >> 1) >> 3) * 32 + ((q) & 1) * 2]
    // where this lane's accumulator elements go in the hidden tile: column chunk*32 + li of rows ... + (e & 3) + 8 (e >> 2) + 4 lh
    int hoff[4];
#pragma unroll
    for (int m = 0; m < 4; m++) hoff[m] = ((((chunk * 32 + li) >> 2) ^ (m + 4 * lh)) << 2) + (li & 3);
    // residual rows of this wave's output chunk (accumulator layout); rows past M read as zero
    const osk::rsrc_t rcx = osk::make_rsrc(a.X, (uint32_t)(a.M * (size_t)D * 4));
    const uint32_t vox = (uint32_t)((4 * lh) * D + chunk * 32 + li) * 4u;
    f32x16v xres[2];
    if (a.Wpp) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            if (rb >= NRB) {
#pragma unroll
                for (int e = 0; e < 16; e++) xres[rb][e] = 0.f;
                continue;
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(row0 + rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2)) * (uint32_t)(D * 4));
                xres[rb][e] = osk::buf_load(rcx, vox, so);
            }
        }
    }
    if (a.lnw && threadIdx.x < 64) {                                      // LayerNorm parameters -> the (still unused) hidden tile
        const int c4 = 4 * (threadIdx.x & 31);
        const float4 t = *reinterpret_cast<const float4 *>((threadIdx.x < 32 ? a.lnw : a.lnb) + c4);
        *reinterpret_cast<float4 *>(&Hs[(threadIdx.x < 32 ? 0 : D) + c4]) = t;
    }
    __syncthreads();                                                      // vmcnt(0) + barrier: the input tile has landed
    if (a.Wpp) {
        // x_new = x + y Wp^T + bp for this wave's 64 rows x 32 columns, kept in registers until the final store; then the
        // tile is replaced by x_new (same swizzled layout) for the LayerNorm below
        constexpr int D8 = 8;
        float wbf[D8], abf[D8][2];
        const osk::rsrc_t rw = osk::make_rsrc(a.Wpp + (size_t)chunk * (D / 2) * 64, (uint32_t)(D / 2) * 256u);
        const float bv = a.bp[chunk * 32 + li];
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int e = 0; e < 16; e++) xres[rb][e] += bv;
#pragma unroll
        for (int d = 0; d < D8; d++) {
            if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, 0u);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(As, d, rb);
        }
#pragma unroll
        for (int q = 0; q < D / 2; q++) {
            const int d = q & (D8 - 1);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) if (rb < NRB) xres[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], xres[rb], 0, 0, 0);
            if (q + D8 < D / 2) {
                if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)(q + D8) * 256u, 0u);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(As, q + D8, rb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                                  // everyone is done reading the attention tile
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            if (rb >= NRB) continue;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                As[r * D + hoff[e & 3]] = xres[rb][e];
            }
        }
        __syncthreads();
    }
    if (a.lnw) {
        // LayerNorm of the tile in place (layernorm_kernel's two passes: mean, then the centred second moment; eps 1e-5): four
        // threads per row, each the eight stored 16-byte slots 8 qd .. 8 qd + 7 (stored slot s holds source slot s ^ (row & 7));
        // the quarter sums meet through two quad permutes
        const int row = threadIdx.x >> 2, qd = threadIdx.x & 3;
        float *rp = As + row * D + qd * 32;
        float4 v4[8];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { v4[j] = *reinterpret_cast<const float4 *>(rp + 4 * j); sum += (v4[j].x + v4[j].y) + (v4[j].z + v4[j].w); }
        sum += dpp_mov<0xB1>(sum); sum += dpp_mov<0x4E>(sum);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v4[j].x -= mean; v4[j].y -= mean; v4[j].z -= mean; v4[j].w -= mean;
            q += (v4[j].x * v4[j].x + v4[j].y * v4[j].y) + (v4[j].z * v4[j].z + v4[j].w * v4[j].w);
        }
        q += dpp_mov<0xB1>(q); q += dpp_mov<0x4E>(q);
        const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int col = 4 * ((8 * qd + j) ^ (row & 7));
            const float4 w4 = *reinterpret_cast<const float4 *>(&Hs[col]), b4 = *reinterpret_cast<const float4 *>(&Hs[D + col]);
            float4 o;
            o.x = v4[j].x * rstd * w4.x + b4.x; o.y = v4[j].y * rstd * w4.y + b4.y;
            o.z = v4[j].z * rstd * w4.z + b4.z; o.w = v4[j].w * rstd * w4.w + b4.w;
            *reinterpret_cast<float4 *>(rp + 4 * j) = o;
        }
        __syncthreads();
    }
    const int nsl = a.Mh / 128;
    for (int s = 0; s < nsl; s++) {
        constexpr int D8 = 8;
        float wbf[D8], abf[D8][2];
        // ---- fc1 slice: hidden columns s*128 + chunk*32 .. +31 for this wave's 64 rows ----
        f32x16v acc1[2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc1[rb][e] = 0.f;
        {
            const osk::rsrc_t rw = osk::make_rsrc(a.W1p + (size_t)(s * 4 + chunk) * (D / 2) * 64, (uint32_t)(D / 2) * 256u);
#pragma unroll
            for (int d = 0; d < D8; d++) {
                if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, 0u);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(As, d, rb);
            }
#pragma unroll
            for (int q = 0; q < D / 2; q++) {
                const int d = q & (D8 - 1);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) if (rb < NRB) acc1[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], acc1[rb], 0, 0, 0);
                if (q + D8 < D / 2) {
                    if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)(q + D8) * 256u, 0u);
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(As, q + D8, rb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            const float bv = a.b1[s * 128 + chunk * 32 + li];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                if (rb >= NRB) continue;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const f2v g = gelu_erf2((f2v){acc1[rb][e] + bv, acc1[rb][e + 1] + bv});
                    Hs[r * D + hoff[e & 3]] = g.x;
                    Hs[(r + 1) * D + hoff[(e + 1) & 3]] = g.y;
                }
            }
        }
        __syncthreads();                                                  // the hidden slice is complete
        // ---- fc2 K-slice s: output columns chunk*32 .. +31 ----
        {
            const osk::rsrc_t rw = osk::make_rsrc(a.W2p + ((size_t)chunk * (a.Mh / 2) + (size_t)s * (D / 2)) * 64, (uint32_t)(D / 2) * 256u);
#pragma unroll
            for (int d = 0; d < D8; d++) {
                if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, 0u);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(Hs, d, rb);
            }
#pragma unroll
            for (int q = 0; q < D / 2; q++) {
                const int d = q & (D8 - 1);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) if (rb < NRB) acc2[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], acc2[rb], 0, 0, 0);
                if (q + D8 < D / 2) {
                    if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)(q + D8) * 256u, 0u);
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG(Hs, q + D8, rb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s + 1 < nsl) __syncthreads();                                 // everyone is done reading the hidden slice
    }
#undef MLP_AFRAG
#define MLP_AFRAG2(T, q, rb) (T)[lo8[((q) >> 1) & 7] + (rb) * 32 * D + (((q) >> 1) >> 3) * 32 + ((q) & 1) * 2]
    // x += acc + b2 (with the projection: x = x_new + acc + b2, no read): rows past M fall outside the descriptor's range
    const float bv = a.b2[chunk * 32 + li];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        if (rb >= NRB) continue;                 // rows of another tile: never stored from here
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(row0 + rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2)) * (uint32_t)(D * 4));
            acc2[rb][e] += bv + (a.Wpp ? xres[rb][e] : osk::buf_load(rcx, vox, so));
            osk::buf_store(rcx, vox, so, acc2[rb][e]);
        }
    }
    if (a.Ynext) {
        // the next block's LayerNorm on the finished rows: tile -> LDS (both tiles are free after the barrier), four threads per
        // row as above, normalised rows straight to Ynext
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            if (rb >= NRB) continue;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                As[r * D + hoff[e & 3]] = acc2[rb][e];
            }
        }
        if (threadIdx.x < 64) {
            const int c4 = 4 * (threadIdx.x & 31);
            *reinterpret_cast<float4 *>(&Hs[(threadIdx.x < 32 ? 0 : D) + c4]) =
                *reinterpret_cast<const float4 *>((threadIdx.x < 32 ? a.lnw_next : a.lnb_next) + c4);
        }
        __syncthreads();
        const int row = threadIdx.x >> 2, qd = threadIdx.x & 3;
        const float *rp = As + row * D + qd * 32;
        float4 v4[8];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { v4[j] = *reinterpret_cast<const float4 *>(rp + 4 * j); sum += (v4[j].x + v4[j].y) + (v4[j].z + v4[j].w); }
        sum += dpp_mov<0xB1>(sum); sum += dpp_mov<0x4E>(sum);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v4[j].x -= mean; v4[j].y -= mean; v4[j].z -= mean; v4[j].w -= mean;
            q += (v4[j].x * v4[j].x + v4[j].y * v4[j].y) + (v4[j].z * v4[j].z + v4[j].w * v4[j].w);
        }
        q += dpp_mov<0xB1>(q); q += dpp_mov<0x4E>(q);
        const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
        if (a.Wqp || (row < nrows && row0 + row < a.M)) {
            float *yp = a.Wqp ? As + row * D + qd * 32 : a.Ynext + (row0 + row) * D;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int col = 4 * ((8 * qd + j) ^ (row & 7));
                const float4 w4 = *reinterpret_cast<const float4 *>(&Hs[col]), b4 = *reinterpret_cast<const float4 *>(&Hs[D + col]);
                float4 o;
                o.x = v4[j].x * rstd * w4.x + b4.x; o.y = v4[j].y * rstd * w4.y + b4.y;
                o.z = v4[j].z * rstd * w4.z + b4.z; o.w = v4[j].w * rstd * w4.w + b4.w;
                *reinterpret_cast<float4 *>(a.Wqp ? yp + 4 * j : yp + col) = o;          // back into the swizzled tile, or out
            }
        }
        if (a.Wqp) {
            // qkv of the next block: three passes of four 32-column chunks over the normalised tile
            __syncthreads();
            const osk::rsrc_t rq = osk::make_rsrc(a.QKVnext, (uint32_t)(a.M * (size_t)(3 * D) * 4));
            for (int p = 0; p < 3; p++) {
                constexpr int D8 = 8;
                float wbf[D8], abf[D8][2];
                const int c = p * 4 + chunk;
                const osk::rsrc_t rw = osk::make_rsrc(a.Wqp + (size_t)c * (D / 2) * 64, (uint32_t)(D / 2) * 256u);
                f32x16v aq[2];
                const float bq = a.bq[c * 32 + li];
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int e = 0; e < 16; e++) aq[rb][e] = bq;
#pragma unroll
                for (int d = 0; d < D8; d++) {
                    if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)d * 256u, 0u);
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG2(As, d, rb);
                }
#pragma unroll
                for (int qq = 0; qq < D / 2; qq++) {
                    const int d = qq & (D8 - 1);
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) if (rb < NRB) aq[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(abf[d][rb], wbf[d], aq[rb], 0, 0, 0);
                    if (qq + D8 < D / 2) {
                        if (NRB > 0) wbf[d] = osk::buf_load(rw, wl + (uint32_t)(qq + D8) * 256u, 0u);
#pragma unroll
                        for (int rb = 0; rb < 2; rb++) if (rb < NRB) abf[d][rb] = MLP_AFRAG2(As, qq + D8, rb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint32_t voq = (uint32_t)((4 * lh) * 3 * D + c * 32 + li) * 4u;
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    if (rb >= NRB) continue;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(row0 + rbp * 64 + rb * 32 + (e & 3) + 8 * (e >> 2)) * (uint32_t)(3 * D * 4));
                        osk::buf_store(rq, voq, so, aq[rb][e]);
                    }
                }
            }
        }
    }
}


template <int BM>
__device__ __forceinline__ void vit_mlp_dispatch(const MlpArgs &a)
{
    const int rbp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >> 2;
    const bool fullt = (int)blockIdx.x < a.nfull;
    const size_t row0 = fullt ? (size_t)blockIdx.x * BM : (size_t)a.nfull * BM + (size_t)((int)blockIdx.x - a.nfull) * a.small_rows;
    const int nrows = fullt ? BM : a.small_rows;                        // rows of this tile (rows past M are masked as before)
    const int left = nrows - rbp * 64;                                  // rows of the tile in this wave's 64-row half
    if (left > 32) vit_mlp_tile<2, BM>(a, row0, nrows);
    else if (left > 0) vit_mlp_tile<1, BM>(a, row0, nrows);
    else vit_mlp_tile<0, BM>(a, row0, nrows);
}
__global__ __launch_bounds__(512, 1) void vit_mlp_kernel(const MlpArgs a) { vit_mlp_dispatch<128>(a); }
__global__ __launch_bounds__(256, 2) void vit_mlp_kernel_bm64(const MlpArgs a) { vit_mlp_dispatch<64>(a); }

// row 0 of every frame: cls token + pos[0]   (transformer_model.py:119-123)
__global__ void cls_row_kernel(int N, int L, int D, const float *cls, const float *pos, float *X)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * D) X[(size_t)(i / D) * L * D + (i % D)] = cls[i % D] + pos[i % D];
}

// latent[n] = sigmoid(LN(X[n][0]))   (transformer_model.py:127-133)
__global__ void cls_head_kernel(int N, int L, int D, const float *X, const float *w, const float *b, float *latent)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *x = X + (size_t)n * L * D;
    float v[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; v[j] = c < D ? x[c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; const float t = c < D ? v[j] - mean : 0.f; q += t * t; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = lane + 64 * j;
        if (c < D) latent[(size_t)n * D + c] = 1.0f / (1.0f + expf(-((v[j] - mean) * rstd * w[c] + b[c])));
    }
}

}  // namespace osv

using namespace osv;

struct os_vit_state {
    VitDims d;
    const float *w;          // caller-owned flat weights
    float *wp; size_t wp_floats;       // every projection matrix re-packed into B-fragment order (vit_pack_w_kernel)
    float *buf; size_t buf_floats;
    bool att_attr_set, mlp_attr_set;
};

static size_t vit_param_count(const VitDims &d)
{
    const size_t D = d.dim, PP = (size_t)d.patch * d.patch, L = ntok(d), Mh = d.mlp;
    size_t n = D * PP + D + D + L * D;
    n += (size_t)d.depth * (2 * D + 3 * D * D + 3 * D + D * D + D + 2 * D + Mh * D + Mh + D * Mh + D);
    return n + 2 * D;
}

void os_vit_destroy(os_ctx *ctx)
{
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v) return;
    if (v->buf) (void)hipFree(v->buf);
    if (v->wp) (void)hipFree(v->wp);
    free(v);
    ctx->vit = nullptr;
}

template <int SRC, int EPI>
static void launch_gemm(os_ctx *ctx, const GemmArgs &g, hipStream_t s, const char *name)
{
    dim3 grid((g.N + 127) / 128, (unsigned)((g.M + GM_BM - 1) / GM_BM));
    const int slot = os_prof_begin(ctx, OS_PHASE_VIT_GEMM, s, name);
    hipLaunchKernelGGL((vit_gemm_kernel<SRC, EPI>), grid, dim3(256), 0, s, g);
    os_prof_end(ctx, slot, s);
}

extern "C" {

size_t os_vit_param_count(const os_vit_dims *d)
{
    VitDims v{d->img_size, d->patch_size, d->embed_dim, d->depth, d->num_heads, d->mlp_hidden};
    return vit_param_count(v);
}

int os_vit_load(os_ctx *ctx, const os_vit_dims *d, const float *w_flat)
{
    OS_CHECK_CTX(ctx);
    if (!d || !w_flat) return os_fail(ctx, -2, "os_vit_load: null pointer");
    if (d->embed_dim > 256 || d->embed_dim % d->num_heads || d->img_size % d->patch_size || d->in_chans != 1)
        return os_fail(ctx, -4, "os_vit_load: unsupported dimensions (embed_dim <= 256, one input channel)");
    const int hd = d->embed_dim / d->num_heads;
    if (hd != 32 && hd != 64) return os_fail(ctx, -4, "os_vit_load: head dimension must be 32 or 64");
    if (d->embed_dim % 128 || d->mlp_hidden % 128 || (d->patch_size * d->patch_size) % 128 || d->patch_size % 4)
        return os_fail(ctx, -4, "os_vit_load: embed_dim, mlp_hidden and patch_size^2 must be multiples of 128 (GEMM k-chunk)");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v) {
        v = (os_vit_state *)calloc(1, sizeof(os_vit_state));
        if (!v) return os_fail(ctx, -13, "os_vit_load: out of memory");
        ctx->vit = v;
    }
    v->d = VitDims{d->img_size, d->patch_size, d->embed_dim, d->depth, d->num_heads, d->mlp_hidden};
    v->w = w_flat;
    // re-pack the projection matrices once (inference weights are static): [patch | per block: qkv, proj, fc1, fc2]
    const size_t D = d->embed_dim, PP = (size_t)d->patch_size * d->patch_size, L = ntok(v->d), Mh = d->mlp_hidden;
    const size_t total = D * PP + (size_t)d->depth * (3 * D * D + D * D + 2 * Mh * D);
    if (os_ensure_scratch(ctx, &v->wp, &v->wp_floats, total)) return -10;
    auto pack = [&](int N, int K, const float *W, float *dst) {
        hipLaunchKernelGGL(vit_pack_w_kernel, dim3(256), dim3(256), 0, 0, N, K, W, dst);
    };
    const float *w = w_flat;
    float *o = v->wp;
    pack((int)D, (int)PP, w, o); o += D * PP; w += D * PP + D + D + L * D;
    for (int blk = 0; blk < d->depth; blk++) {
        w += 2 * D;
        pack((int)(3 * D), (int)D, w, o); o += 3 * D * D; w += 3 * D * D + 3 * D;
        pack((int)D, (int)D, w, o); o += D * D; w += D * D + D;
        w += 2 * D;
        pack((int)Mh, (int)D, w, o); o += Mh * D; w += Mh * D + Mh;
        pack((int)D, (int)Mh, w, o); o += D * Mh; w += D * Mh + D;
    }
    OS_HIP(ctx, hipGetLastError());
    OS_HIP(ctx, hipStreamSynchronize(0));
    return 0;
}

int os_vit_encode(os_ctx *ctx, int32_t N, const float *images, float *latent, void *stream)
{
    OS_CHECK_CTX(ctx);
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v || !v->w) return os_fail(ctx, -5, "os_vit_encode: call os_vit_load first");
    if (N <= 0 || !images || !latent) return os_fail(ctx, -2, "os_vit_encode: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const VitDims &d = v->d;
    const int D = d.dim, L = ntok(d), G = grid_of(d), PP = d.patch * d.patch, Mh = d.mlp, hd = D / d.heads;
    const size_t M = (size_t)N * L, Mp = (size_t)N * G * G;
    // scratch: X [M][D] (token matrix) | Y [M][D] | big [M][max(3D, mlp)]
    const size_t bigw = (size_t)(3 * D > Mh ? 3 * D : Mh);
    const size_t need = M * D * 2 + M * bigw;
    if (os_ensure_scratch(ctx, &v->buf, &v->buf_floats, need)) return -10;
    float *X = v->buf, *Y = X + M * D, *big = Y + M * D;
    const float *w = v->w;
    const float *wp = v->wp;
    const float *patch_b = w + (size_t)D * PP, *cls = patch_b + D, *pos = cls + D;
    w = pos + (size_t)L * D;
    GemmArgs g;
    g.img = images; g.img_size = d.img; g.grid = G; g.pos = pos; g.L = L; g.A = nullptr; g.lda = 0;
    // patch embedding straight from the images into the token matrix (+ patch bias + position table); cls rows beside it
    g.M = Mp; g.N = D; g.K = PP; g.Wp = wp; g.bias = patch_b; g.C = X;
    g.patch = d.patch;
    launch_gemm<1, 3>(ctx, g, s, "vit_gemm_kernel<patch,+pos>");
    wp += (size_t)D * PP;
    {
        const int slot = os_prof_begin(ctx, OS_PHASE_VIT_MISC, s, "cls_row_kernel");
        hipLaunchKernelGGL(cls_row_kernel, dim3((N * D + 255) / 256), dim3(256), 0, s, N, L, D, cls, pos, X);
        os_prof_end(ctx, slot, s);
    }
    OS_HIP(ctx, hipGetLastError());
    g.M = M;
    for (int blk = 0; blk < d.depth; blk++) {
        const float *ln1w = w; w += D; const float *ln1b = w; w += D;
        w += (size_t)3 * D * D; const float *qkvb = w; w += 3 * D;
        w += (size_t)D * D; const float *projb = w; w += D;
        const float *ln2w = w; w += D; const float *ln2b = w; w += D;
        w += (size_t)Mh * D; const float *fc1b = w; w += Mh;
        w += (size_t)D * Mh; const float *fc2b = w; w += D;
        int slot;
        const bool tail_fused = D == 128 && ctx->tune_vit_mlp_fused >= 2;
        if (!(tail_fused && blk > 0)) {                  // blocks after the first get their LayerNorm from the previous block's tail kernel
            slot = os_prof_begin(ctx, OS_PHASE_VIT_MISC, s, "layernorm_kernel");
            hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, M, D, X, ln1w, ln1b, Y, (size_t)D);
            os_prof_end(ctx, slot, s);
        }
        if (!(tail_fused && blk > 0 && ctx->tune_vit_mlp_fused >= 3)) {     // later blocks: qkv already produced by the previous tail kernel
            g.A = Y; g.lda = D; g.N = 3 * D; g.K = D; g.Wp = wp; g.bias = qkvb; g.C = big;
            launch_gemm<0, 0>(ctx, g, s, "vit_gemm_kernel<+bias>");
        }
        wp += (size_t)3 * D * D;
        const size_t alds = (size_t)2 * L * (hd + 1) * sizeof(float);
        const int ntl = (L + 31) / 32;
        slot = os_prof_begin(ctx, OS_PHASE_VIT_ATTN, s, (hd == 32 && ntl == 7) ? (ctx->tune_vit_att_dma ? "attention_mfma_dma_kernel<7>" : "attention_mfma_kernel<7>") : "attention_kernel");
        if (hd == 32 && ntl == 7) {              // the reference's shape: 197 tokens, head_dim 32 -> matrix cores
            const size_t mlds = (size_t)2 * L * 33 * sizeof(float);
            if (!v->att_attr_set) {
                OS_HIP(ctx, hipFuncSetAttribute((const void *)attention_mfma_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)attention_mfma_dma_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                v->att_attr_set = true;
            }
            const int nitems = N * d.heads, cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
            if (ctx->tune_vit_att_dma) {          // persistent workgroups, K / V of the next head by LDS-DMA underneath the current one
                const size_t dlds = (size_t)2 * L * (36 + 32) * sizeof(float);
                hipLaunchKernelGGL(attention_mfma_dma_kernel<7>, dim3(nitems < cus ? nitems : cus), dim3(512), dlds, s, L, D, d.heads, nitems, big, Y);
            } else
            hipLaunchKernelGGL(attention_mfma_kernel<7>, dim3(N * d.heads), dim3(512), mlds, s, L, D, d.heads, big, qkvb, Y);
        } else if (hd == 32) hipLaunchKernelGGL(attention_kernel<32>, dim3(N * d.heads), dim3(256), alds, s, L, D, d.heads, big, qkvb, Y);
        else hipLaunchKernelGGL(attention_kernel<64>, dim3(N * d.heads), dim3(256), alds, s, L, D, d.heads, big, qkvb, Y);
        os_prof_end(ctx, slot, s);
        const bool mlp_fused = D == 128 && ctx->tune_vit_mlp_fused;
        const float *wproj = wp;
        if (!(mlp_fused && ctx->tune_vit_mlp_fused >= 2)) {
            g.A = Y; g.lda = D; g.N = D; g.K = D; g.Wp = wp; g.bias = projb; g.C = X;
            launch_gemm<0, 2>(ctx, g, s, "vit_gemm_kernel<+bias,+residual>");
        }
        wp += (size_t)D * D;
        if (!mlp_fused) {
            slot = os_prof_begin(ctx, OS_PHASE_VIT_MISC, s, "layernorm_kernel");
            hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, M, D, X, ln2w, ln2b, Y, (size_t)D);
            os_prof_end(ctx, slot, s);
        }
        if (mlp_fused) {
            // LayerNorm + fc1 + GELU + fc2 + residual in one kernel: neither the normalised rows nor the hidden activations
            // leave the CU
            MlpArgs ma;
            ma.M = M; ma.Mh = Mh; ma.Y = X; ma.lnw = ln2w; ma.lnb = ln2b; ma.X = X; ma.Wpp = nullptr; ma.bp = nullptr;
            ma.W1p = wp; ma.b1 = fc1b; ma.W2p = wp + (size_t)Mh * D; ma.b2 = fc2b;
            ma.lnw_next = ma.lnb_next = nullptr; ma.Ynext = nullptr;
            ma.Wqp = nullptr; ma.bq = nullptr; ma.QKVnext = nullptr;
            if (tail_fused && blk + 1 < d.depth) {           // the next block's ln1 parameters follow this block's in the flat vector
                ma.lnw_next = fc2b + D; ma.lnb_next = fc2b + 2 * D; ma.Ynext = Y;
                if (ctx->tune_vit_mlp_fused >= 3) {          // ... and its qkv weights / bias: [ln1w ln1b | Wqkv (3D x D) | bqkv]
                    ma.Wqp = wp + (size_t)2 * Mh * D; ma.bq = fc2b + 3 * D + (size_t)3 * D * D; ma.QKVnext = big;
                }
            }
            if (ctx->tune_vit_mlp_fused >= 2) { ma.Y = Y; ma.Wpp = wproj; ma.bp = projb; }      // attention output -> projection -> LN2 -> MLP in one pass
            if (!v->mlp_attr_set) {
                OS_HIP(ctx, hipFuncSetAttribute((const void *)vit_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)vit_mlp_kernel_bm64, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                v->mlp_attr_set = true;
            }
            slot = os_prof_begin(ctx, OS_PHASE_VIT_GEMM, s, ctx->tune_vit_mlp_bm == 64 ? "vit_mlp_kernel_bm64" : "vit_mlp_kernel");
            // tiles: whole rounds of full tiles over the workgroup slots, then the remainder as 32- or 64-row tiles when those fit
            // one more (short) round -- otherwise the last round runs a full tile's time on a fraction of the chip
            const int TB = ctx->tune_vit_mlp_bm == 64 ? 64 : GM_BM;
            const size_t ntiles = (M + TB - 1) / TB;
            const size_t cu = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * (TB == 64 ? 2 : 1);      // workgroup slots
            size_t nfull = ntiles, nsmall = 0;
            int small_rows = TB;
            const size_t rem = ntiles % cu;
            if (rem != 0 && ctx->tune_vit_tail_split) {
                const size_t rows_rem = M - (ntiles - rem) * TB;
                small_rows = (rows_rem + 31) / 32 <= cu ? 32 : ((rows_rem + 63) / 64 <= cu ? 64 : TB);
                if (small_rows < TB) { nfull = ntiles - rem; nsmall = (rows_rem + small_rows - 1) / small_rows; }
                else small_rows = TB;
            }
            ma.nfull = (int)nfull; ma.small_rows = small_rows;
            if (TB == 64) hipLaunchKernelGGL(vit_mlp_kernel_bm64, dim3((unsigned)(nfull + nsmall)), dim3(256), (size_t)2 * 64 * 128 * sizeof(float), s, ma);
            else hipLaunchKernelGGL(vit_mlp_kernel, dim3((unsigned)(nfull + nsmall)), dim3(512), (size_t)2 * GM_BM * 128 * sizeof(float), s, ma);
            os_prof_end(ctx, slot, s);
            wp += (size_t)2 * Mh * D;
        } else {
            g.A = Y; g.lda = D; g.N = Mh; g.K = D; g.Wp = wp; g.bias = fc1b; g.C = big;
            launch_gemm<0, 1>(ctx, g, s, "vit_gemm_kernel<+bias,gelu>");
            wp += (size_t)Mh * D;
            g.A = big; g.lda = Mh; g.N = D; g.K = Mh; g.Wp = wp; g.bias = fc2b; g.C = X;
            launch_gemm<0, 2>(ctx, g, s, "vit_gemm_kernel<+bias,+residual>");
            wp += (size_t)D * Mh;
        }
        OS_HIP(ctx, hipGetLastError());
    }
    const float *nw = w; w += D; const float *nb = w;
    const int slot = os_prof_begin(ctx, OS_PHASE_VIT_MISC, s, "cls_head_kernel");
    hipLaunchKernelGGL(cls_head_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, L, D, X, nw, nb, latent);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
