// Semantics probe (development aid): v_permlane16_swap through the builtin (ROCm 7.2 drops its second result) and the DPP
// row_half_mirror / row_mirror controls used by the ViT attention softmax reductions.
#include <hip/hip_runtime.h>
__global__ void k(float *o)
{
    float v = (float)threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    o[threadIdx.x] = __builtin_bit_cast(float, r[0]); o[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
    int d = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false);
    o[128 + threadIdx.x] = __builtin_bit_cast(float, d);
    d = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false);
    o[192 + threadIdx.x] = __builtin_bit_cast(float, d);
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 4; s++) { for (int i = 0; i < 64; i++) printf("%g ", h[64 * s + i]); printf("\n"); }
    return 0;
}
