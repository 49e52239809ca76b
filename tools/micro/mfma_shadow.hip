// How much of a VALU instruction's cost disappears when it is issued right behind a v_mfma_f32_32x32x2_f32 by the SAME wave
// (one wave per SIMD)?  Per loop iteration: 4 x [one MFMA (own accumulator), then NV independent VALU instructions of one kind].
// Reported: cycles per iteration-quarter (MFMA + NV VALU) at the measured clock, beside the same NV instructions without the MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP> __device__ __forceinline__ void valu(float &x, f2 &p, float &acc_a)
{
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(1.0001f));
    if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 3) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc_a) : "v"(x));
    if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
}

template <int NV, int OP, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    float d[16]; f2 pp[16]; float aa[16];
    for (int i = 0; i < 16; i++) { d[i] = a + i; pp[i] = (f2){a + i, a - i}; aa[i] = 0.f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; v++) valu<OP>(d[v & 15], pp[v & 15], aa[v & 15]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    for (int i = 0; i < 16; i++) s += d[i] + pp[i][0] + pp[i][1];
    if (OP == 3) for (int i = 0; i < 16; i++) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(aa[i])); s += t; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int OP, bool MFMA> float run(float *d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, OP, MFMA>), dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, OP, MFMA>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int OP> void table(float *d, const char *name, double ghz)
{
    const int it = 10000;
    const double c = ghz * 1e6 / (it * 4.0);            // ms -> cycles per (MFMA + NV VALU)
    printf("%-18s with MFMA : NV=0 %.1f  1 %.1f  2 %.1f  4 %.1f  8 %.1f  12 %.1f  16 %.1f  24 %.1f  32 %.1f cycles\n", name,
           run<0, OP, true>(d, it) * c, run<1, OP, true>(d, it) * c, run<2, OP, true>(d, it) * c, run<4, OP, true>(d, it) * c, run<8, OP, true>(d, it) * c,
           run<12, OP, true>(d, it) * c, run<16, OP, true>(d, it) * c, run<24, OP, true>(d, it) * c, run<32, OP, true>(d, it) * c);
    printf("%-18s VALU only : NV=4 %.1f  8 %.1f  16 %.1f  32 %.1f cycles\n", name,
           run<4, OP, false>(d, it) * c, run<8, OP, false>(d, it) * c, run<16, OP, false>(d, it) * c, run<32, OP, false>(d, it) * c);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    // clock: MFMA-only stream, 64 cycles per instruction plus issue
    const float ms0 = run<0, 0, true>(d, 10000);
    const double ghz = 2.4;
    printf("MFMA only: %.3f ms for 40000 MFMAs per wave = %.1f cycles each at %.1f GHz nominal\n", ms0, ms0 * ghz * 1e6 / 40000, ghz);
    table<0>(d, "v_fma_f32", ghz);
    table<1>(d, "v_pk_fma_f32", ghz);
    table<4>(d, "v_pk_add_f32", ghz);
    table<2>(d, "v_exp_f32", ghz);
    table<3>(d, "v_accvgpr_write", ghz);
    return 0;
}
