#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o)
{
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]); printf("\n");
    printf("r1:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]); printf("\n");
    return 0;
}
