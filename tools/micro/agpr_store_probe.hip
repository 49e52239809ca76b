// Micro-probe (development): buffer_store_dword with an AGPR data operand, right behind the v_accvgpr_write and much later.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(float *p, int n, int delay)
{
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, n * 4, 0x00020000);
    float a, b;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"((float)threadIdx.x + 0.5f));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b) : "v"((float)threadIdx.x + 1000.5f));
    uint32_t vo = threadIdx.x * 4, so0 = 0, so1 = 256;
    if (delay) __builtin_amdgcn_s_sleep(20);
    asm volatile("buffer_store_dword %0, %1, %2, %3 offen" : : "a"(a), "v"(vo), "s"(r), "s"(so0) : "memory");
    asm volatile("buffer_store_dword %0, %1, %2, %3 offen" : : "a"(b), "v"(vo), "s"(r), "s"(so1) : "memory");
}
int main()
{
    float *d, h[128];
    hipMalloc(&d, sizeof(h));
    int bad = 0;
    for (int delay = 0; delay < 2; delay++) {
        hipMemset(d, 0, sizeof(h));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 128, delay);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) if (h[i] != i + 0.5f || h[64 + i] != i + 1000.5f) bad++;
        printf("delay %d: h[0] %g h[63] %g h[64] %g h[127] %g\n", delay, h[0], h[63], h[64], h[127]);
    }
    printf(bad ? "AGPR store: WRONG (%d)\n" : "AGPR store: ok\n", bad);
    return bad != 0;
}
