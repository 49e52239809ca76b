// gru_device.hpp -- device helpers shared by the GRU layer kernels (gru_kernels.hip) and the split-bf16 layer kernel
// (gru_bf16_kernels.hip): the packed cell update and the LDS-DMA / inline-asm LDS accessors of the staged kernels.
#pragma once
#include "gru_common.hpp"
#include "kf_device.hpp"

namespace osg {

using osk::rsrc_t;

// Development build only (-DOS_LAYER_TS, tools/layer_ts.sh): shader-clock stamps at the phase boundaries of a step, summed over
// the steps by thread 0 of workgroup 0 and printed (gru_layer_kernel and gru_layer_stage_kernel).
#ifdef OS_LAYER_TS
#define OSL_TS_DECL unsigned long long ts_prev = 0, ts_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define OSL_TS(i)                                                                  \
    {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                         \
        const unsigned long long now = __builtin_readcyclecounter();               \
        if ((i) > 0) ts_sum[i] += now - ts_prev;                                   \
        ts_prev = now;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }
#else
#define OSL_TS_DECL
#define OSL_TS(i)
#endif

// The GRU cell (gru/gru_model.py:16; gate order r | z | n) for TWO elements held in adjacent accumulator registers: the eleven
// non-transcendental operations of an element are v_pk_* instructions on the pair, the six transcendentals stay scalar.
// a_r, a_z: W_i. x + W_h. h pre-activations without bias; a_n = W_in x; a_h = W_hn h; nb_*: biases pre-scaled for exp2
// (-log2 e (b_ir + b_hr), -log2 e (b_iz + b_hz), 2 log2 e b_in), b_hn as it is.  Same IEEE operations as the scalar form.
struct CellPair { osk::f2 r, z, n, ghn, hn; };
__device__ __forceinline__ CellPair gru_cell_pair(osk::f2 a_r, osk::f2 a_z, osk::f2 a_n, osk::f2 a_h, osk::f2 hprev, float nb_r, float nb_z,
                                                  float nb_n, float b_hn)
{
    using osk::f2; using osk::fma2; using osk::splat2;
    constexpr float LOG2E = 1.44269504088896341f;
    CellPair c;
    const f2 tr = fma2(a_r, splat2(-LOG2E), splat2(nb_r));
    const f2 tz = fma2(a_z, splat2(-LOG2E), splat2(nb_z));
    const f2 dr = (f2){__builtin_amdgcn_exp2f(tr[0]), __builtin_amdgcn_exp2f(tr[1])} + splat2(1.0f);
    const f2 dz = (f2){__builtin_amdgcn_exp2f(tz[0]), __builtin_amdgcn_exp2f(tz[1])} + splat2(1.0f);
    c.r = (f2){__builtin_amdgcn_rcpf(dr[0]), __builtin_amdgcn_rcpf(dr[1])};
    c.z = (f2){__builtin_amdgcn_rcpf(dz[0]), __builtin_amdgcn_rcpf(dz[1])};
    c.ghn = a_h + splat2(b_hn);
    const f2 u = fma2(c.r, c.ghn, a_n);
    const f2 tn = fma2(u, splat2(2.0f * LOG2E), splat2(nb_n));
    const f2 dn = (f2){__builtin_amdgcn_exp2f(tn[0]), __builtin_amdgcn_exp2f(tn[1])} + splat2(1.0f);
    c.n = fma2(splat2(-2.0f), (f2){__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])}, splat2(1.0f));
    c.hn = fma2(c.z, hprev - c.n, c.n);                 // (1 - z) n + z h
    return c;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (non-template helpers: inside a kernel template hipcc's host pass drops the launch stub when it meets these builtins)
__device__ __forceinline__ void stage_dma16(rsrc_t r, float *l, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, 0);
}
// seq_out stores of the stage kernel: DEFAULT cache policy, not non-temporal.  A 64-byte line of the SoA sequence (16 trajectories of
// one hidden unit) is completed by four 16-byte stores of two lane halves within a few instructions; streamed (nt) they reached HBM as
// partial sectors -- 4.88 GB written per launch against 2.52 GB algorithmic -- while the write-back L2 merges them: 2.525 GB, the
// kernel's traffic 958 B per (trajectory, step) = 1.00x algorithmic, and 10.2 -> 9.9 ms per layer (profiles/r04_traffic_ref_shape.txt).
__device__ __forceinline__ void buf_store4(rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
template <int OFF>
__device__ __forceinline__ float lds_read_asm(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write_asm(uint32_t addr, float v)
{
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

}  // namespace osg
