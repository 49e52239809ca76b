// gru_common.hpp -- types and helpers shared by the GRU layer kernels and the fused Kalman+GRU kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace osg {

struct LayerArgs {
    int B, T, K, H;            // K = input width of this layer
    int KPx, KPh;              // k-pairs of the x part (ceil(K/2)) and of the h part (H/2)
    const float *xs;           // [T][K][B], or the caller's (B, T, K) batch_first tensor when xs_btf (gru_layer_ahead_kernel only)
    int xs_btf = 0;
    const float *w;            // packed weights of this layer (see pack kernel)
    float *seq_out;            // [T][H][B] or null
    float *h_last;             // [H][B] or null
    // training only (null for inference): per-step activations saved row-major [T][B][H] for the backward sweep
    float *sv_r, *sv_z, *sv_n, *sv_g, *sv_h;   // r, z, n, gh_n (= W_hn h + b_hn), h_t
    // window-stream inference (os_gru_forward_windows, first layer only; null otherwise): gi [gi_rows][3H] = rows . W_ih^T of the
    // ROW stream, computed once per row; trajectory b of this launch is the window starting at row b, so its step t takes row b + t
    const float *gi = nullptr;
    int gi_rows = 0;
    // opt-in split-bf16 gate GEMM (gru_layer_bf16_kernel; null otherwise): this layer's weights as bf16 terms in the fragment
    // order of v_mfma_f32_32x32x16_bf16, KBx = 16-wide k-blocks of the x part (rounded up to even)
    const uint32_t *wbf = nullptr;
    int KBx = 0;
};

// gru_wide_kernel (gru_wide_kernel.hip): the layer stack of a small batch on four CUs per (layer, tile)
struct WideArgs {
    int n, tiles, B, T, K0;               // layers in this launch, 32-row tiles, batch, steps, input width of layer 0
    const float *xs0;                     // layer 0's input: SoA [T][K0][B], or the caller's batch_first (B, T, K0) tensor when xs0_btf
    int xs0_btf;
    const float *w[8];                    // packed weight image of each layer (os_gru_load)
    float *hseq[8];                       // h stream of each layer, row-major [T][B][H]: the exchange buffer (training: the saved h stream)
    float *sv_r[8], *sv_z[8], *sv_n[8], *sv_g[8];      // SAVE: the other saved activations, [T][B][H]
    float *h_last[8];                     // optional h_T of a layer, SoA [H][B]
    uint32_t *flags;                      // [n][tiles][4] progress counters: steps published
    int32_t *err, *err_local;
    uint32_t max_polls;
    int drop_layer, drop_step;            // tests: that layer stops publishing from that step on
};

__host__ __device__ inline size_t chunk_floats(int KPx, int KPh) { return (size_t)(KPx + KPh) * 3 * 64 + 4 * 32; }

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each):
// absolute error < 2e-7 on outputs in (0,1) / (-1,1), far inside the 1e-5 parity bar of the GRU head.
__device__ __forceinline__ float sigmoidf_(float v)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}
__device__ __forceinline__ float tanhf_(float v)
{
    // tanh(v) = 1 - 2 / (1 + e^{2v}); e^{2v} -> inf gives 1, -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * v));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0));
// in the T-step loops of the layer / sweep kernels each phase ends with dozens of write-once global stores per thread that no
// thread of the workgroup ever reads back, and waiting for them to land cost 1-2 us per barrier (write latency), twice per step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- split-bf16 operands (opt-in reduced-precision gate GEMMs: fused_kf_gru_bf16_kernel, gru_layer_bf16_kernel) ----
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float a, float b)       // (low half = a, high half = b), round to nearest even
{
    const bf16x2_t v = __builtin_convertvector((f2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo_f32(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi_f32(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// one pair of fp32 values -> SPL dwords of packed bf16 terms (hi [, mid], lo); the remainders are exact in fp32
template <int SPL>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t *t)
{
    t[0] = pack_bf16(a, b);
    float ra = a - bf16_lo_f32(t[0]), rb = b - bf16_hi_f32(t[0]);
    if (SPL == 3) {
        t[1] = pack_bf16(ra, rb);
        ra -= bf16_lo_f32(t[1]); rb -= bf16_hi_f32(t[1]);
    }
    t[SPL - 1] = pack_bf16(ra, rb);
}

// Launch-wide latch of the progress-counter kernels (round 6): once ANY workgroup of the context has given a wait up, the others stop
// waiting too -- a workgroup that starts late reads the word before its first wait, a waiting one looks at it every 1,024 polls.  Without
// this a starved launch paid the bounded wait once per ROUND of late workgroups (measured: 5.7 s for 14 rounds of 0.67 s on two free
// CUs per XCD, tests/test_gpu_contention.py); with it the whole launch costs one wait.
__device__ __forceinline__ bool stack_lost_already(const int32_t *err_local)
{
    return err_local && __hip_atomic_load(err_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

}  // namespace osg
