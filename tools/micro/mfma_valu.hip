// Does VALU fp32 work co-execute with v_mfma_f32_32x32x2_f32 issued by the SAME wave (one wave per SIMD)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, bool BF16>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    float d[8];
    for (int i = 0; i < 8; i++) d[i] = a + i;
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 ab; for (int i = 0; i < 8; i++) ab[i] = (short)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (BF16) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m], 0, 0, 0);
            else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) d[v & 7] = fmaf(d[v & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    for (int i = 0; i < 8; i++) s += d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, bool BF16> float run(float *d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, BF16>), dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, BF16>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    const int it = 20000;   // 80k MFMAs per wave
    printf("fp32 32x32x2 : NV=0 %.3f ms  NV=4 %.3f  NV=8 %.3f  NV=12 %.3f  NV=16 %.3f  (ideal MFMA-only at 2.4GHz: %.3f ms)\n",
           run<0, false>(d, it), run<4, false>(d, it), run<8, false>(d, it), run<12, false>(d, it), run<16, false>(d, it), it * 4 * 64 / 2.4e6);
    printf("bf16 32x32x16: NV=0 %.3f ms  NV=2 %.3f  NV=4 %.3f  NV=6 %.3f  NV=8 %.3f  (ideal: %.3f ms)\n",
           run<0, true>(d, it), run<2, true>(d, it), run<4, true>(d, it), run<6, true>(d, it), run<8, true>(d, it), it * 4 * 32 / 2.4e6);
    return 0;
}
