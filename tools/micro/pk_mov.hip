// Semantics probe for v_pk_mov_b32 op_sel on gfx950 (development aid): prints D = (lo, hi) for every op_sel / op_sel_hi setting.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out)
{
    f2 a = {1.f, 2.f}, b = {3.f, 4.f}, d;
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b)); out[0] = d[0]; out[1] = d[1];
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b)); out[2] = d[0]; out[3] = d[1];
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[4] = d[0]; out[5] = d[1];
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b)); out[6] = d[0]; out[7] = d[1];
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b)); out[8] = d[0]; out[9] = d[1];
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); out[10] = d[0]; out[11] = d[1];
}
int main()
{
    float *d, h[12];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *n[6] = {"op_sel:[0,0]", "op_sel:[1,0]", "op_sel:[0,1]", "op_sel:[1,1]", "[0,0] hi:[0,0]", "[0,0] hi:[1,1]"};
    for (int i = 0; i < 6; i++) printf("a=(1,2) b=(3,4) %s -> (%g, %g)\n", n[i], h[2 * i], h[2 * i + 1]);
    return 0;
}
