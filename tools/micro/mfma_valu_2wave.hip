// Two waves per SIMD: do a VALU-only wave and an fp32-MFMA-only wave on the SAME SIMD overlap (time = max) or share
// the fp32 pipe (time = sum)?  512-thread workgroups, one per CU: waves 0-3 and 4-7 land pairwise on the 4 SIMDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int MODE, bool BF16>   // MODE 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, waves 4-7 VALU
__global__ __launch_bounds__(512, 1) void k(float *out, int it_m, int it_v)
{
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || (MODE == 2 && wave >= 4);
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0001f, d[8];
    for (int i = 0; i < 8; i++) d[i] = a + i;
    s8 ab; for (int i = 0; i < 8; i++) ab[i] = (short)(threadIdx.x + i);
    if (do_m)
        for (int it = 0; it < it_m; it++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (BF16) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            }
    if (do_v)
        for (int it = 0; it < it_v; it++)
#pragma unroll
            for (int v = 0; v < 64; v++) d[v & 7] = fmaf(d[v & 7], 1.0001f, 0.5f);
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    for (int i = 0; i < 8; i++) s += d[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, bool BF16> float run(float *d, int im, int iv)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(256), dim3(512), 0, 0, d, 10, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(256), dim3(512), 0, 0, d, im, iv);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int im = 10000, iv = 10000;     // 40k MFMAs per MFMA wave; 640k FMAs per VALU wave
    printf("fp32 MFMA : 8 waves MFMA %.3f ms | 8 waves VALU %.3f ms | 4 MFMA + 4 VALU waves (half the work of each) %.3f ms\n",
           run<0, false>(d, im, iv), run<1, false>(d, im, iv), run<2, false>(d, im, iv));
    printf("bf16 MFMA : 8 waves MFMA %.3f ms | 8 waves VALU %.3f ms | 4 MFMA + 4 VALU waves %.3f ms\n",
           run<0, true>(d, im, iv), run<1, true>(d, im, iv), run<2, true>(d, im, iv));
    return 0;
}
